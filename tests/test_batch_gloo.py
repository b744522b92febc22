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


# ---- intra-proof sharding with the REAL prover: two ranks share GPU 0, the all-gather of the partial commitments goes through
# gloo (zkfhe_comm_create_with_transport); on an 8-GPU node the same library path runs over RCCL (zkfhe_comm_create).
def _sharded_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch  # noqa: F401
    import zk_fhe_amd as zk
    from oracle import circuit_ref as C
    from tests.test_proof_oracle import synth_input
    ctx = zk.Context(0)
    prm = C.BfvParams(N=8)
    inp = json.dumps(synth_input(8, prm.Q, prm.T, prm.B, 1))
    cfg = zk.bfv_auto_config(inp, (8, prm.Q, prm.T, prm.B), 9, unusable_rows=9)
    comm = zk.Comm(ctx, rank, world, all_gather=lambda b: B.all_gather_bytes(b, world))
    lo, hi = comm.point_range(512)
    srs = zk.Srs(ctx, 9, comm=comm)
    pk = zk.BfvProvingKey(ctx, srs, inp, (8, prm.Q, prm.T, prm.B), cfg)
    proof, inst, _ = pk.prove(inp, b"shard")
    info = pk.info()
    # a k = 13 column batch through zkfhe_msm_batch_sharded directly as well: 5 columns x 8192 over the two point ranges
    res = {"rank": rank, "range": (lo, hi), "proof": proof, "inst": list(inst), "vk": info["vk_digest"]}
    if rank == 0:
        # the unsharded reference on the same GPU, same seed
        srs1 = zk.Srs(ctx, 9)
        pk1 = zk.BfvProvingKey(ctx, srs1, inp, (8, prm.Q, prm.T, prm.B), cfg)
        res["proof_1gpu"], _, _ = pk1.prove(inp, b"shard")
        res["vk_1gpu"] = pk1.info()["vk_digest"]
        pk1.destroy()
        srs1.destroy()
    out.put(res)
    dist.barrier()
    pk.destroy()
    srs.destroy()
    comm.destroy()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_prover_two_ranks_one_gpu_same_bytes():
    """zkfhe_srs_create_sharded + zkfhe_bfv_keygen / zkfhe_bfv_prove over a 2-rank communicator: every commitment is the
    sum of two point-range partials gathered across processes; verifying key and proof must equal the single-GPU ones."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = sorted([q.get(timeout=600), q.get(timeout=600)], key=lambda r: r["rank"])
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got[0]["range"] == (0, 256) and got[1]["range"] == (256, 512)
    assert got[0]["proof"] == got[1]["proof"] and got[0]["inst"] == got[1]["inst"]          # all ranks hold the same proof
    assert got[0]["vk"] == got[0]["vk_1gpu"] and got[0]["proof"] == got[0]["proof_1gpu"]    # ... the single-GPU one
