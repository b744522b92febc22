"""The N > 1 path (independent proofs sharded over ranks, gather of proof bytes, max-over-ranks timing) with two
gloo processes on CPU.  The prover is replaced by a stand-in that hashes its input: the sharding / gathering /
concurrency logic is what is under test here; the GPU prover itself is covered by the -m gpu tests."""
import hashlib
import json
import os
import socket

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import zk_fhe_amd.batch as B


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class FakeKey:
    def prove(self, text, seed, ctx=None):
        return hashlib.sha256(text.encode() + seed + str(ctx).encode()[:0]).digest(), [], [0] * 5


def _worker(rank, world, port, n_items, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    inputs = ["input-%d" % i for i in range(n_items)]
    seeds = [b"seed-%d" % i for i in range(n_items)]
    mine = B.shard_indices(n_items, rank, world)
    proofs = B.prove_batch(FakeKey(), [inputs[i] for i in mine], [seeds[i] for i in mine], contexts=["s0", "s1", "s2"])
    allp = B.gather_proofs(dict(zip(mine, proofs)), n_items, rank, world)
    t = B.max_over_ranks(1.0 + rank)
    if rank == 0:
        out.put((allp, t))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_batch_matches_single_process():
    n_items, world = 11, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in ps:
        p.start()
    allp, t = q.get(timeout=120)
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = [hashlib.sha256(("input-%d" % i).encode() + b"seed-%d" % i).digest() for i in range(n_items)]
    assert allp == want            # every proof present, in order, identical to the 1-rank result
    assert t == 2.0                # max over ranks


def test_shard_indices_cover_everything_once():
    for world in (1, 2, 4, 8):
        seen = sorted(i for r in range(world) for i in B.shard_indices(64, r, world))
        assert seen == list(range(64))
        assert all(len(B.shard_indices(64, r, world)) == 64 // world for r in range(world))


def test_run_concurrent_propagates_errors():
    import pytest

    def fn(w, j):
        if j == 3:
            raise ValueError("boom")
        return j
    with pytest.raises(ValueError):
        B.run_concurrent(list(range(8)), ["a", "b"], fn)


def _gather_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    local = bytes([rank]) * 64 * 3          # three 64-byte "points" per rank
    parts = B.all_gather_bytes(local, world)
    if rank == 1:
        out.put(parts)
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_bytes_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    parts = q.get(timeout=120)
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert parts == [bytes([0]) * 192, bytes([1]) * 192]


def test_point_ranges_partition():
    for n, w in ((524288, 8), (8192, 3), (10, 4)):
        rs = [B.point_range(n, r, w) for r in range(w)]
        assert rs[0][0] == 0 and rs[-1][1] == n
        assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))


# ---- intra-proof sharding with the REAL prover: the ranks share GPU 0, the all-gather of the partial commitments goes through
# gloo (zkfhe_comm_create_with_transport); on an 8-GPU node the same library path runs over RCCL (zkfhe_comm_create).
HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden", "bfv")


def _circuit(which):
    """(keygen input text, proving input text, (N, Q, T, B), config maker) for "toy" (k = 9, N = 8) and "bfv13" (the reference's
    data/bfv/bfv.in at k = 13 with the pinned configs/bfv.json layout)."""
    import zk_fhe_amd as zk
    from oracle import circuit_ref as C
    if which == "toy":
        from tests.test_proof_oracle import synth_input
        prm = C.BfvParams(N=8)
        inp = json.dumps(synth_input(8, prm.Q, prm.T, prm.B, 1))
        cfg = zk.bfv_auto_config(inp, (8, prm.Q, prm.T, prm.B), 9, unusable_rows=9)
        return inp, inp, (8, prm.Q, prm.T, prm.B), cfg, 9
    if which == "k14":   # rows longer than one NTT tile: the long-row transforms on every rank's column share
        from tests.test_proof_oracle import synth_input
        prm = C.BfvParams(N=16)
        inp = json.dumps(synth_input(16, prm.Q, prm.T, prm.B, 5))
        cfg = zk.bfv_auto_config(inp, (16, prm.Q, prm.T, prm.B), 14, unusable_rows=109)
        return inp, inp, (16, prm.Q, prm.T, prm.B), cfg, 14
    prm = C.BfvParams()
    cfgj = json.load(open(os.path.join(G, "bfv_config.json")))
    return (open(os.path.join(G, "bfv_empty.in")).read(), open(os.path.join(G, "bfv.in")).read(), (1024, prm.Q, prm.T, prm.B),
            zk.BfvConfig.from_pinning(cfgj), 13)


def _sharded_worker(rank, world, port, which, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("ZKFHE_TABLE_GB", "4" if world <= 4 else "1")   # per SRS half and per rank: eight ranks share this one GPU
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch  # noqa: F401
    import zk_fhe_amd as zk
    ctx = zk.Context(0)
    text_kg, text, params, cfg, k = _circuit(which)
    comm = zk.Comm(ctx, rank, world, all_gather=lambda b: B.all_gather_bytes(b, world))
    lo, hi = comm.point_range(1 << k)
    srs = zk.Srs(ctx, k, comm=comm)
    pk = zk.BfvProvingKey(ctx, srs, text_kg, params, cfg)
    proof, inst, _ = pk.prove(text, b"shard")
    info = pk.info()
    res = {"rank": rank, "range": (lo, hi), "proof": proof, "inst": list(inst), "vk": info["vk_digest"], "table_bits": srs.table_bits()}
    if rank == 0:
        # the unsharded reference on the same GPU, same seed
        srs1 = zk.Srs(ctx, k)
        pk1 = zk.BfvProvingKey(ctx, srs1, text_kg, params, cfg)
        res["proof_1gpu"], _, _ = pk1.prove(text, b"shard")
        res["vk_1gpu"] = pk1.info()["vk_digest"]
        res["table_bits_1gpu"] = srs1.table_bits()
        ok, why = zk.bfv_verify(pk1.export_vk(), inst, proof)
        res["verified"] = (ok, why)
        pk1.destroy()
        srs1.destroy()
    out.put(res)
    dist.barrier()
    pk.destroy()
    srs.destroy()
    comm.destroy()
    ctx.close()
    dist.destroy_process_group()


def _run_ranks(target, world, *args, timeout=900):
    """Runs target(rank, world, port, *args, queue) in `world` spawned processes and returns their results by rank.  A rank that dies
    (a failed assertion, an out-of-memory kill) would leave the others waiting in a collective until gloo's own timeout: the
    survivors are terminated and the test fails at once, with the exit codes."""
    import queue as queue_mod
    import time
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=target, args=(r, world, port) + args + (q,)) for r in range(world)]
    for p in ps:
        p.start()
    got, deadline = [], time.time() + timeout
    try:
        while len(got) < world:
            try:
                got.append(q.get(timeout=2))
            except queue_mod.Empty:
                dead = [(r, p.exitcode) for r, p in enumerate(ps) if p.exitcode not in (None, 0)]
                assert not dead, "rank(s) exited early (rank, exit code): %s" % dead
                assert time.time() < deadline, "ranks did not finish within %d s" % timeout
        for p in ps:
            p.join(timeout=120)
            assert p.exitcode == 0
    finally:
        for p in ps:
            if p.is_alive():
                p.terminate()
        for p in ps:
            p.join(timeout=10)
            if p.is_alive():
                p.kill()
    return sorted(got, key=lambda r: r["rank"])


@pytest.mark.gpu
@pytest.mark.parametrize("which,world", [("toy", 2), ("toy", 3), ("toy", 8), ("bfv13", 2), ("bfv13", 4), ("bfv13", 8), ("k14", 2)])
def test_sharded_prover_ranks_share_one_gpu_same_bytes(which, world):
    """zkfhe_srs_create_sharded + zkfhe_bfv_keygen / zkfhe_bfv_prove over a W-rank communicator: every commitment is the sum of
    W point-range partials gathered across processes, and the coset extension + quotient are sharded by column (each rank
    extends and evaluates only the columns of its permutation chunks; the W partial quotients are gathered and summed);
    verifying key and proof must equal the single-GPU ones.  Three ranks: ragged chunk ranges; eight ranks (SURVEY.md section 4:
    byte identity at 1 / 2 / 4 / 8; BASELINE configs[4] itself -- ONE k = 19 proof over eight ranks against the single-GPU bytes -- is
    tests/test_bench_ranks.py::test_bench_one_k19_proof_sharded_over_eight_ranks_gloo, through bench.py): on the toy circuit more ranks than permutation chunks, so some ranks own no column.  "bfv13" is the
    reference's bfv.in at k = 13: each rank's SRS slice takes the digit-multiple table path (k_msm_table) with its own,
    wider digits (a slice of 2^13 / W points fits more bits into the same budget than the whole basis)."""
    got = _run_ranks(_sharded_worker, world, which)
    n = {"toy": 512, "bfv13": 8192, "k14": 16384}[which]
    assert [g["range"] for g in got] == [(n * r // world, n * (r + 1) // world) for r in range(world)]
    for g in got[1:]:
        assert g["proof"] == got[0]["proof"] and g["inst"] == got[0]["inst"] and g["vk"] == got[0]["vk"]   # all ranks hold the same proof
    assert got[0]["vk"] == got[0]["vk_1gpu"] and got[0]["proof"] == got[0]["proof_1gpu"]                  # ... the single-GPU one
    assert got[0]["verified"][0], got[0]["verified"][1]
    if which == "bfv13":
        assert all(g["table_bits"][0] >= got[0]["table_bits_1gpu"][0] >= 8 for g in got)   # the table path ran on every slice


def _msm_worker(rank, world, port, log_n, n_cols, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("ZKFHE_TABLE_GB", "4")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np
    import torch  # noqa: F401
    import zk_fhe_amd as zk
    from oracle import binding as orc
    from oracle import pyref
    n = 1 << log_n
    ctx = zk.Context(0)
    rng = np.random.default_rng(99 + log_n)
    bases = orc.g1_powers(orc.ints_to_mont([4242])[0], orc.ints_to_mont([0x7654321])[0], n)
    bases[n // 3] = 0
    S = np.frombuffer(rng.bytes(n_cols * n * 32), dtype=np.uint64).reshape(n_cols, n, 4).copy()
    S[..., 3] &= (1 << 60) - 1       # below r: valid (non-canonical Montgomery representatives are still field elements)
    S[1] = orc.ints_to_mont([int(v) for v in rng.integers(0, 256, n)])   # a witness-like short column
    S[2, : n // 2] = 0
    comm = zk.Comm(ctx, rank, world, all_gather=lambda b: B.all_gather_bytes(b, world))
    lo, hi = comm.point_range(n)
    slice_basis = zk.Basis(ctx, bases[lo:hi])
    got = ctx.msm_sharded(comm, slice_basis, S, lo)
    res = {"rank": rank, "got": got.tobytes(), "slice_table": slice_basis.has_table}
    if rank == 0:
        full = zk.Basis(ctx, bases)
        res["single"] = ctx.msm(full, S).tobytes()
        res["oracle"] = orc.msm(S[:2], bases).tobytes()
        full.destroy()
    out.put(res)
    dist.barrier()
    slice_basis.destroy()
    comm.destroy()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,world,n_cols", [(13, 2, 5), (13, 4, 20), (18, 2, 3)])
def test_msm_batch_sharded_directly(log_n, world, n_cols):
    """zkfhe_msm_batch_sharded on its own: k = 13 column batches (table path on every slice; 20 columns = the 256-partials-per-visit
    fold) and a k = 18 batch (BASELINE config 5 regime: no table exists for a slice of 2^17 points under the suite's budget,
    so every rank runs the bucket pipeline on its point range).  All ranks must hold the single-GPU commitments; two columns are
    also checked against the CPU oracle."""
    got = _run_ranks(_msm_worker, world, log_n, n_cols, timeout=1500)
    for g in got:
        assert g["got"] == got[0]["single"]
    assert got[0]["single"][: 2 * 64] == got[0]["oracle"]
    assert all(g["slice_table"] == (log_n <= 13) for g in got)


@pytest.mark.gpu
def test_rccl_transport_one_rank_smoke():
    """The RCCL branch of csrc/comm.hip on ONE GPU: zkfhe_comm_unique_id (ncclGetUniqueId), zkfhe_comm_create with world = 1
    and an id (a real ncclCommInitRank), zkfhe_comm_all_gather (ncclAllGather of ncclUint8 on the context's stream) and a whole
    zkfhe_msm_batch_sharded through it -- the dlopen'ed entry points, their prototypes (taken from rccl.h at build time) and the
    datatype constant all execute.  The multi-rank topology itself needs the 8-GPU node."""
    import numpy as np
    import torch  # noqa: F401
    import zk_fhe_amd as zk
    from oracle import binding as orc
    ctx = zk.Context(0)
    uid = zk.Comm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = zk.Comm(ctx, 0, 1, unique_id=uid)
    rng = np.random.default_rng(5)
    payload = np.frombuffer(rng.bytes(64 * 37 + 5), dtype=np.uint8)     # not a multiple of anything: ncclUint8 counts bytes
    send, recv = ctx.to_device(payload), ctx.alloc(payload.nbytes)
    comm.all_gather(send, recv, payload.nbytes)
    ctx.sync()
    assert np.array_equal(recv.download(dtype=np.uint8), payload)
    # the asynchronous form (the prover's quotient shares, one coset row at a time): three gathers in a row on the communicator's own
    # stream, each behind the kernel that produced its input on the context's stream, then one join
    rows = orc.ints_to_mont([int.from_bytes(rng.bytes(31), "little") for _ in range(3 * 4096)]).reshape(3, 4096, 4)
    src, tmp, dst = ctx.to_device(rows), ctx.alloc(rows.nbytes), ctx.alloc(rows.nbytes)
    row_bytes = rows.nbytes // 3
    for k1 in range(3):
        ctx.fr_binop_dev("add", src.at(k1 * row_bytes), src.at(k1 * row_bytes), tmp.at(k1 * row_bytes), 4096)   # produce row k1 ...
        comm.all_gather_async(tmp, dst, row_bytes, k1 * row_bytes, k1 * row_bytes)                                                 # ... and send it
    comm.join()
    ctx.sync()
    assert np.array_equal(dst.download(shape=(3, 4096, 4)), orc.fe_binop("add", rows.reshape(-1, 4), rows.reshape(-1, 4)).reshape(3, 4096, 4))
    src.free(), tmp.free(), dst.free()
    n, n_cols = 1024, 3
    bases = orc.g1_powers(orc.ints_to_mont([77])[0], orc.ints_to_mont([99])[0], n)
    S = orc.ints_to_mont([int.from_bytes(rng.bytes(31), "little") for _ in range(n_cols * n)]).reshape(n_cols, n, 4)
    basis = zk.Basis(ctx, bases)
    assert np.array_equal(ctx.msm_sharded(comm, basis, S, 0), orc.msm(S, bases))
    basis.destroy()
    send.free(), recv.free()
    comm.destroy()
    ctx.close()


@pytest.mark.gpu
def test_proof_through_one_rank_rccl_communicator_matches_plain():
    """The whole collective path of the sharded prover on ONE GPU: an SRS created with a one-rank RCCL communicator
    (zkfhe_comm_create, world = 1, a real unique id) makes every commitment of keygen and prove go through
    zkfhe_msm_batch_sharded_async -- the partial MSM on the context's stream, ncclAllGather(ncclUint8) and the sum on the
    communicator's own stream, two gather buffers alternating, the phase-0 points returned through an event recorded behind the
    collective while the gadget kernels already run -- and the key and the proof bytes must be the plain single-GPU ones.
    Both transcripts: with Poseidon the early phase-1 commitment puts two commitment batches in flight at once."""
    import torch  # noqa: F401
    import zk_fhe_amd as zk
    ctx = zk.Context(0)
    try:
        uid = zk.Comm.unique_id()
    except zk.ZkfheError as e:
        pytest.skip("librccl.so is not loadable here: %s" % e)
    comm = zk.Comm(ctx, 0, 1, unique_id=uid)
    for which in ("toy", "bfv13"):
        text_kg, text, params, cfg, k = _circuit(which)
        for transcript in ("poseidon", "blake2b"):
            cfg_t = zk.BfvConfig(cfg.k, cfg.n_gate0, cfg.n_gate1, cfg.n_lookup, cfg.n_rlc, cfg.unusable_rows, cfg.lookup_bits, cfg.break_points, transcript)
            out = []
            for c in (comm, None):
                srs = zk.Srs(ctx, k, comm=c)
                pk = zk.BfvProvingKey(ctx, srs, text_kg, params, cfg_t)
                proofs = [pk.prove(text, b"rccl-%d" % i)[0] for i in range(3)]     # three in a row: the gather buffers alternate
                out.append((pk.info()["vk_digest"], proofs))
                pk.destroy()
                srs.destroy()
            assert out[0] == out[1], (which, transcript)
    comm.destroy()
    ctx.close()


# ---- BASELINE configs[2]: a batch of 64 independent k = 13 proofs through batch.shard_indices / prove_batch / gather_proofs with
# the REAL prover.  Eight gloo ranks sharing GPU 0 (the shape of the 8-GPU run: proof i -> rank i mod 8, no data-path collective,
# the proofs gathered on every rank), and one rank with 16 streams.  The 64 inputs are SURVEY.md 8(d) config 3's: the
# reference's bfv.in and 63 seeded vectors, every one under its own public key.
def _batch64_worker(rank, world, port, n_streams, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("ZKFHE_TABLE_GB", "1" if world > 1 else "4")   # per SRS half and per rank: the ranks share this one GPU
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch  # noqa: F401
    import zk_fhe_amd as zk
    from zk_fhe_amd import inputs as gen
    text_kg, text, params, cfg, k = _circuit("bfv13")
    inputs = gen.config3_batch(text, 64)
    seeds = [b"batch64-%d" % i for i in range(64)]
    ctxs = [zk.Context(0) for _ in range(n_streams)]
    host = B.configure_host(zk, world)
    srs = zk.Srs(ctxs[0], k)
    pk = zk.BfvProvingKey(ctxs[0], srs, text_kg, params, cfg)
    vk = pk.export_vk()
    mine = B.shard_indices(64, rank, world)
    got = B.prove_batch(pk, [inputs[i] for i in mine], [seeds[i] for i in mine], ctxs, with_instances=True)
    verdicts = [zk.bfv_verify(vk, inst, proof) for proof, inst in got]            # every proof of this rank: zkfhe_bfv_verify (pairing check)
    allp = B.gather_proofs({i: p for i, (p, _) in zip(mine, got)}, 64, rank, world)
    res = {"rank": rank, "mine": mine, "verified": [v[0] for v in verdicts], "why": [v[1] for v in verdicts if not v[0]],
           "inst_len": [len(inst) for _, inst in got], "host": host, "all": allp if rank == 0 else None, "n_all": len(allp),
           "all_sha": hashlib.sha256(b"".join(allp)).hexdigest()}
    out.put(res)
    dist.barrier()
    pk.destroy()
    srs.destroy()
    for c in ctxs:
        c.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_batch_of_64_eight_ranks_equals_one_rank_and_cpu_prover():
    """configs[2] on the hardware there is: 64 distinct encryptions, proof i -> rank i mod 8 over eight processes sharing GPU 0
    (two streams each), against ONE process with 16 streams.  Every proof passes zkfhe_bfv_verify on the rank that made it; rank 0's
    gathered list is complete and in input order, identical on all ranks, identical between the 8-rank and the 1-rank run; and
    four sampled proofs (bfv.in's and three synthetic ones, from different ranks) equal the native CPU prover's bytes
    (oracle/cpu_prover.cpp with the oracle's key and SRS)."""
    from oracle import cpu_prover as CP
    from oracle import halo2_ref as H
    from tests.test_gpu_prover import oracle_k13
    from zk_fhe_amd import inputs as gen
    got8 = _run_ranks(_batch64_worker, 8, 2, timeout=1500)
    got1 = _run_ranks(_batch64_worker, 1, 16, timeout=1500)
    for got, world in ((got8, 8), (got1, 1)):
        assert sorted(i for g in got for i in g["mine"]) == list(range(64))
        for g in got:
            assert g["mine"] == list(range(g["rank"], 64, world))
            assert all(g["verified"]), g["why"][:2]
            assert all(n == 5121 for n in g["inst_len"])                                   # pk0, pk1, c0, c1, cyclo (examples/bfv.rs:118-122)
            assert g["n_all"] == 64 and g["all_sha"] == got[0]["all_sha"]                  # every rank holds the same gathered list
    assert got8[0]["host"]["hash_mode"] == "shared" or got8[0]["host"]["cpus_per_rank"] >= B.CPUS_PER_GPU_FOR_LATENCY_MODE
    all8, all1 = got8[0]["all"], got1[0]["all"]
    assert all8 == all1                                                                     # same bytes whichever rank / stream made a proof
    assert len(set(all8)) == 64                                                             # 64 different proofs
    o = oracle_k13()
    hcfg = H.Config.from_pinning(o["cfgj"])
    cp = CP.CpuProver(hcfg, o["pk_o"], o["srs_o"], o["prm"])
    inputs = gen.config3_batch(o["text"], 64)
    for i in (0, 13, 38, 63):                                                               # ranks 0, 5, 6, 7 of the 8-rank run
        assert cp.prove(inputs[i].decode(), b"batch64-%d" % i) == all8[i], "proof %d differs from the CPU prover's" % i
    cp.close()
    # ... and to the Python oracle's own proofs of the same inputs and seeds (an independent prover; made once on the CPU by
    # tests/golden/gen_batch64_digests.py, committed as digests)
    gold = json.load(open(os.path.join(HERE, "golden", "batch64_proofs.json")))
    assert gold["count"] == 64 and set(gold["proofs"]) == {"0", "13", "38", "63"}
    for i, g in gold["proofs"].items():
        assert hashlib.sha256(inputs[int(i)]).hexdigest() == g["input_sha256"]
        assert hashlib.sha256(all8[int(i)]).hexdigest() == g["proof_sha256"], "proof %s differs from the oracle prover's" % i

