"""GPU prover (zkfhe_bfv_keygen / zkfhe_bfv_prove) vs the CPU oracle prover: byte-identical proofs for the same
seed, and the oracle verifier (real pairing check) accepts them.  Run on the MI355X box: pytest -m gpu."""
import json
import os

import pytest

from oracle import circuit_ref as C
from oracle import halo2_ref as H
from tests.test_proof_oracle import synth_input

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden", "bfv")


@pytest.fixture(scope="module")
def ctx():
    import torch  # noqa: F401
    import zk_fhe_amd as zk
    c = zk.Context(0)
    yield c
    c.close()


def first_diff(a, b):
    for i in range(0, min(len(a), len(b)), 32):
        if a[i:i + 32] != b[i:i + 32]:
            return i // 32
    return None if len(a) == len(b) else min(len(a), len(b)) // 32


@pytest.mark.parametrize("transcript", ["poseidon", "blake2b"])
def test_toy_proof_bytes_match_oracle(ctx, transcript):
    """Both transcripts: snark-verifier's Poseidon (what the reference proves with, examples/bfv.rs:311) and halo2's Blake2b."""
    import zk_fhe_amd as zk
    prm = C.BfvParams(N=8)
    inp = synth_input(8, prm.Q, prm.T, prm.B, 1)
    circ = H.BfvCircuit(inp, prm)
    hcfg = H.auto_config(9, 9, circ, transcript=transcript)
    srs_o = H.make_srs(9)
    pk_o, _ = H.keygen_circuit(hcfg, circ, srs_o)
    proof_o, inst_o = H.prove(hcfg, pk_o, srs_o, circ, b"seed-1")
    assert H.verify(H.VerifyingKey(pk_o), srs_o, inst_o, proof_o)
    srs = zk.Srs(ctx, 9)
    zcfg = zk.BfvConfig(9, hcfg.n_gate0, hcfg.n_gate1, hcfg.n_lookup, hcfg.n_rlc, 9, transcript=transcript)
    pk = zk.BfvProvingKey(ctx, srs, json.dumps(inp), (8, prm.Q, prm.T, prm.B), zcfg)
    info = pk.info()
    assert info["break_points"] == pk_o.break_points
    assert info["fixed_commit"] == pk_o.fixed_commit
    assert info["sigma_commit"] == pk_o.sigma_commit
    assert info["vk_digest"] == pk_o.vk_digest
    proof, inst, tm = pk.prove(json.dumps(inp), b"seed-1")
    assert inst == inst_o
    assert first_diff(proof, proof_o) is None, "first differing 32-byte item: %s" % first_diff(proof, proof_o)
    assert H.verify(H.VerifyingKey(pk_o), srs_o, inst, proof)
    ok, why = zk.bfv_verify(pk.export_vk(), inst, proof)
    assert ok, why
    # a different seed changes the bytes but still verifies; a wrong witness is refused
    proof2, _, _ = pk.prove(json.dumps(inp), b"seed-2")
    assert proof2 != proof and H.verify(H.VerifyingKey(pk_o), srs_o, inst, proof2)
    bad = dict(inp)
    c0 = list(bad["c0"])
    c0[1] = str((int(c0[1]) + 1) % prm.Q)
    bad["c0"] = c0
    with pytest.raises(zk.ZkfheError):
        pk.prove(json.dumps(bad), b"seed-1")
    pk.destroy()
    srs.destroy()


_ORACLE_K13 = {}


def oracle_k13():
    """The CPU oracle's keygen + proof of the reference's bfv.in (seed b"seed-1"): ~40 s, made once per session."""
    if not _ORACLE_K13:
        cfgj = json.load(open(os.path.join(G, "bfv_config.json")))
        hcfg = H.Config.from_pinning(cfgj)
        bp = {"gate0": cfgj["break_points"]["gate"][0], "gate1": cfgj["break_points"]["gate"][1], "rlc": cfgj["break_points"]["rlc"]}
        prm = C.BfvParams()
        text_empty = open(os.path.join(G, "bfv_empty.in")).read()
        text = open(os.path.join(G, "bfv.in")).read()
        srs_o = H.make_srs(13)
        pk_o, _ = H.keygen_circuit(hcfg, H.BfvCircuit(json.loads(text_empty), prm), srs_o, bp)
        proof_o, inst_o = H.prove(hcfg, pk_o, srs_o, H.BfvCircuit(json.loads(text), prm), b"seed-1")
        _ORACLE_K13.update(cfgj=cfgj, bp=bp, prm=prm, text_empty=text_empty, text=text, srs_o=srs_o, pk_o=pk_o, proof_o=proof_o, inst_o=inst_o)
    return _ORACLE_K13


@pytest.mark.parametrize("table_gb", ["4", None, "160"])
def test_bfv_in_k13_proof_bytes_match_oracle(ctx, monkeypatch, table_gb):
    """BASELINE config 2: the reference's data/bfv/bfv.in, k = 13, pinned configs/bfv.json layout, at three table budgets.
    "160" is the SERVICE profile bench.py and the driver's BENCH line measure: 15-bit digits for the Lagrange half (146 GB: more
    than 2^31 table entries), 13-bit for the monomial half (43 GB), every commitment a sum of table points (k_msm_table) -- 14 bits
    if the device does not have the room when the test runs.  None is the library's own DEFAULT (48 GB and at most a quarter of the
    free memory: 13 bits for the Lagrange half, 11 for the monomial one: 56 GB).  "4" is the budget the rest of the suite runs on: 9-bit digits, which serve the calls of a
    few columns only -- the two wide calls take the bucket pipeline (k_msm_accumulate ...).  Between them the cases prove bfv.in
    through both MSM paths and three table geometries."""
    import zk_fhe_amd as zk
    monkeypatch.delenv("ZKFHE_TABLE_BITS", raising=False)
    if table_gb is None:
        monkeypatch.delenv("ZKFHE_TABLE_GB", raising=False)
    else:
        monkeypatch.setenv("ZKFHE_TABLE_GB", table_gb)
    o = oracle_k13()
    cfgj, prm, text_empty, text = o["cfgj"], o["prm"], o["text_empty"], o["text"]
    srs = zk.Srs(ctx, 13)
    bits, wide = srs.table_bits()
    ti = srs.table_info()
    print("tables at ZKFHE_TABLE_GB=%s: %s" % (table_gb, ti))
    if table_gb == "160":
        assert bits in (14, 15) and wide and ti["bits"][1] == bits and ti["gb"] > 100
    elif table_gb is None:
        assert (bits, wide) == (13, True) and ti["bits"] == (11, 13) and 50 < ti["gb"] < 54 and not ti["narrowed"]   # 40 + 12 GiB
    else:
        assert (bits, wide) == (9, False), (bits, wide)
    zcfg = zk.BfvConfig.from_pinning(cfgj)
    zcfg_nobp = zk.BfvConfig(zcfg.k, zcfg.n_gate0, zcfg.n_gate1, zcfg.n_lookup, zcfg.n_rlc, zcfg.unusable_rows, zcfg.lookup_bits)
    pk = zk.BfvProvingKey(ctx, srs, text_empty, (1024, prm.Q, prm.T, prm.B), zcfg_nobp)
    info = pk.info()
    assert info["break_points"] == o["bp"]  # keygen recomputes the reference's 158 pinned break points
    proof, inst, tm = pk.prove(text, b"seed-1")
    print("GPU prove timings [witness, commit, quotient, open, total] ms:", tm)
    assert len(inst) == 5121
    assert info["vk_digest"] == o["pk_o"].vk_digest
    assert H.verify(H.VerifyingKey(o["pk_o"]), o["srs_o"], inst, proof), "oracle verifier (pairing check) rejects the GPU proof"
    ok, why = zk.bfv_verify(pk.export_vk(), inst, proof)   # the product's own C++ verifier (host CPU)
    assert ok, why
    assert inst == o["inst_o"]
    assert first_diff(proof, o["proof_o"]) is None, "first differing 32-byte item: %s" % first_diff(proof, o["proof_o"])
    pk.destroy()
    srs.destroy()


def test_native_cpu_prover_k13_bytes_match_oracle_and_gpu(ctx):
    """bench.py's cpu_baseline leg (oracle/cpu_prover.cpp) on the reference's bfv.in at the pinned k = 13 layout: the same bytes
    as the oracle prover and as the GPU prover."""
    from oracle import cpu_prover as CP
    o = oracle_k13()
    hcfg = H.Config.from_pinning(o["cfgj"])
    cp = CP.CpuProver(hcfg, o["pk_o"], o["srs_o"], o["prm"])
    proof_c = cp.prove(o["text"], b"seed-1")
    cp.close()
    assert proof_c == o["proof_o"]
    import zk_fhe_amd as zk
    srs = zk.Srs(ctx, 13)
    pk = zk.BfvProvingKey(ctx, srs, o["text_empty"], (1024, o["prm"].Q, o["prm"].T, o["prm"].B), zk.BfvConfig.from_pinning(o["cfgj"]))
    proof, _, _ = pk.prove(o["text"], b"seed-1")
    assert proof == proof_c
    pk.destroy()
    srs.destroy()


def test_gpu_gate_stream_matches_oracle_cell_for_cell(ctx):
    """SURVEY.md 8a rows A8-A14 directly: the 1 231 992 phase-1 gate cells the GPU gadget kernels emit for the reference's
    data/bfv/bfv.in equal the oracle's restatement of src/poly_chip.rs + halo2-base (oracle/circuit_ref.py), for a fixed gamma."""
    import numpy as np
    import zk_fhe_amd as zk
    cfgj = json.load(open(os.path.join(G, "bfv_config.json")))
    prm = C.BfvParams()
    inp = C.load_input(os.path.join(G, "bfv.in"))
    gamma = 0x1F3A5C7E9B2D4F60123456789ABCDEF0FEDCBA9876543210
    _, _, st = C.bfv_phase0(inp, prm)
    ctx_gate, _ = C.bfv_phase1(st, prm, gamma)
    want = ctx_gate.advice
    assert len(want) == 1231992
    srs = zk.Srs(ctx, 13)
    pk = zk.BfvProvingKey(ctx, srs, open(os.path.join(G, "bfv_empty.in")).read(), (1024, prm.Q, prm.T, prm.B), zk.BfvConfig.from_pinning(cfgj))
    got = pk.witness_stream(open(os.path.join(G, "bfv.in")).read(), gamma)
    assert got.shape == (len(want), 4)
    mask = (1 << 64) - 1
    ref = np.array([[(v >> (64 * j)) & mask for j in range(4)] for v in want], dtype=np.uint64)
    bad = np.nonzero((got != ref).any(axis=1))[0]
    assert bad.size == 0, "first differing cell: %d" % bad[0]
    pk.destroy()
    srs.destroy()


def test_gpu_gate_stream_60bit_modulus(ctx):
    """Same cell-for-cell check with a 60-bit Q (BASELINE config 4 shape at N = 256): div_mod on 130-bit values, 17-limb range
    checks -- the widths the k = 13 parameters never reach."""
    import numpy as np
    import zk_fhe_amd as zk
    from zk_fhe_amd import inputs as gen
    N, Q, T, B, k = 256, (1 << 60) - 93, 7, 19, 12
    inp = gen.generate(N, Q, T, B, seed=9)
    text = json.dumps(inp)
    prm = C.BfvParams(N=N, Q=Q, T=T, B=B)
    gamma = 0x0123456789ABCDEF0123456789ABCDEF0123456789ABCDEF
    _, _, st = C.bfv_phase0(inp, prm)
    ctx_gate, _ = C.bfv_phase1(st, prm, gamma)
    want = ctx_gate.advice
    probe = zk.bfv_build_tables(text, (N, Q, T, B), zk.BfvConfig(k, 8, 400, 120, 16, 109), 1, keygen_mode=False)
    n0, n1, nr = (len(probe["break_points"][w]) + 1 for w in ("gate0", "gate1", "rlc"))
    nl = -(-probe["lookups"] // ((1 << k) - 109))
    srs = zk.Srs(ctx, k)
    pk = zk.BfvProvingKey(ctx, srs, text, (N, Q, T, B), zk.BfvConfig(k, n0, n1, nl, nr, 109))
    got = pk.witness_stream(text, gamma)
    mask = (1 << 64) - 1
    ref = np.array([[(v >> (64 * j)) & mask for j in range(4)] for v in want], dtype=np.uint64)
    assert got.shape == ref.shape
    bad = np.nonzero((got != ref).any(axis=1))[0]
    assert bad.size == 0, "first differing cell: %d" % bad[0]
    proof, inst, _ = pk.prove(text, b"q60")
    ok, why = zk.bfv_verify(pk.export_vk(), inst, proof)
    assert ok, why
    # and the whole proof byte for byte against the oracle prover: the 130-bit columns go through the same blinding draw
    # order, commitments, quotient and openings as the 29-bit ones
    hcfg = H.Config(k, n0, n1, nl, nr, 109)
    srs_o = H.make_srs(k)
    circ = H.BfvCircuit(inp, prm)
    pk_o, _ = H.keygen_circuit(hcfg, circ, srs_o)
    assert pk.info()["vk_digest"] == pk_o.vk_digest
    proof_o, inst_o = H.prove(hcfg, pk_o, srs_o, circ, b"q60")
    assert inst == inst_o
    assert first_diff(proof, proof_o) is None, "first differing 32-byte item: %s" % first_diff(proof, proof_o)
    pk.destroy()
    srs.destroy()


def test_device_and_host_witness_generators_agree(ctx):
    """The phase-1 gate stream is generated on the GPU by default; ZKFHE_WITNESS=host keeps the host generator.  Same
    seed -> identical bytes, on the reference's input and on a second random one."""
    import zk_fhe_amd as zk
    cfgj = json.load(open(os.path.join(G, "bfv_config.json")))
    prm = C.BfvParams()
    srs = zk.Srs(ctx, 13)
    pk = zk.BfvProvingKey(ctx, srs, open(os.path.join(G, "bfv_empty.in")).read(), (1024, prm.Q, prm.T, prm.B), zk.BfvConfig.from_pinning(cfgj))
    texts = [open(os.path.join(G, "bfv.in")).read(), json.dumps(synth_input(1024, prm.Q, prm.T, prm.B, 77))]
    try:
        for i, text in enumerate(texts):
            os.environ.pop("ZKFHE_WITNESS", None)
            dev, inst_d, _ = pk.prove(text, b"w-%d" % i)
            os.environ["ZKFHE_WITNESS"] = "host"
            host, inst_h, _ = pk.prove(text, b"w-%d" % i)
            assert inst_d == inst_h
            assert first_diff(dev, host) is None, "first differing 32-byte item: %s" % first_diff(dev, host)
            # The default with the Poseidon transcript commits the phase-1 columns before the challenge and adds the
            # challenge-dependent cells as sparse corrections afterwards (ZKFHE_EARLY_P1, prove.hip) -- the same points,
            # hence the same bytes as with the plain order
            os.environ.pop("ZKFHE_WITNESS", None)
            os.environ["ZKFHE_EARLY_P1"] = "0"
            plain, inst_e, _ = pk.prove(text, b"w-%d" % i)
            os.environ.pop("ZKFHE_EARLY_P1", None)
            assert inst_e == inst_d and first_diff(plain, dev) is None
            # The phase-0 / RLC columns reach the device through a kernel that reads the pinned witness table (round 6);
            # ZKFHE_UPLOAD=copy keeps the copy command + in-place conversion it replaced
            os.environ["ZKFHE_UPLOAD"] = "copy"
            copied, inst_c, _ = pk.prove(text, b"w-%d" % i)
            os.environ.pop("ZKFHE_UPLOAD", None)
            assert inst_c == inst_d and first_diff(copied, dev) is None
    finally:
        os.environ.pop("ZKFHE_WITNESS", None)
        os.environ.pop("ZKFHE_EARLY_P1", None)
        os.environ.pop("ZKFHE_UPLOAD", None)
    pk.destroy()
    srs.destroy()


def test_generated_inputs_and_rejected_witnesses(ctx):
    """zk_fhe_amd.inputs (real BFV keygen + encrypt) -> prove -> the C++ verifier accepts; inputs that break a range
    check or the ciphertext identity are refused by the GPU witness path with a status, not proved."""
    import zk_fhe_amd as zk
    from zk_fhe_amd import inputs
    cfgj = json.load(open(os.path.join(G, "bfv_config.json")))
    prm = C.BfvParams()
    srs = zk.Srs(ctx, 13)
    pk = zk.BfvProvingKey(ctx, srs, json.dumps(inputs.empty(1024)), (1024, prm.Q, prm.T, prm.B), zk.BfvConfig.from_pinning(cfgj), replay=True)
    vk = pk.export_vk()
    good = inputs.generate(1024, prm.Q, prm.T, prm.B, seed=5)
    proof, inst, _ = pk.prove(json.dumps(good), b"gen-5")
    ok, why = zk.bfv_verify(vk, inst, proof)
    assert ok, why
    # e0 outside [-B, B]: the range gadget's is_less_than output is 0, the copy to the constant 1 is violated
    bad = dict(good)
    e0 = list(bad["e0"])
    e0[7] = str(prm.B + 1)
    bad["e0"] = e0
    with pytest.raises(zk.ZkfheError):
        pk.prove(json.dumps(bad), b"gen-5")
    # u outside {0, 1, Q-1}
    bad = dict(good)
    u = list(bad["u"])
    u[3] = "2"
    bad["u"] = u
    with pytest.raises(zk.ZkfheError):
        pk.prove(json.dumps(bad), b"gen-5")
    # a coefficient >= Q is refused before any GPU work (src/poly.rs:28 asserts coeff <= modulus; Q itself then fails the field check)
    bad = dict(good)
    c1 = list(bad["c1"])
    c1[0] = str(prm.Q + 5)
    bad["c1"] = c1
    with pytest.raises(zk.ZkfheError):
        pk.prove(json.dumps(bad), b"gen-5")
    # the context is still usable afterwards
    proof2, inst2, _ = pk.prove(json.dumps(good), b"gen-5")
    assert proof2 == proof and inst2 == inst
    pk.destroy()
    srs.destroy()


def test_proving_key_save_and_load(ctx, tmp_path):
    """zkfhe_bfv_pk_save / zkfhe_bfv_pk_load: the reloaded key proves the same bytes; damaged files are refused."""
    import zk_fhe_amd as zk
    cfgj = json.load(open(os.path.join(G, "bfv_config.json")))
    prm = C.BfvParams()
    srs = zk.Srs(ctx, 13)
    pk = zk.BfvProvingKey(ctx, srs, open(os.path.join(G, "bfv_empty.in")).read(), (1024, prm.Q, prm.T, prm.B), zk.BfvConfig.from_pinning(cfgj))
    text = open(os.path.join(G, "bfv.in")).read()
    proof, inst, _ = pk.prove(text, b"pk-io")
    path = str(tmp_path / "bfv.pk")
    pk.save(path)
    info = pk.info()
    pk.destroy()
    pk2 = zk.BfvProvingKey.load(ctx, srs, path, 1024)
    assert pk2.info() == info
    proof2, inst2, _ = pk2.prove(text, b"pk-io")
    assert proof2 == proof and inst2 == inst
    pk2.destroy()
    raw = bytearray(open(path, "rb").read())
    # where the BFV parameters sit: magic | 8 x u32 | three break-point lists (u64 count + u32 each) | N Q T B
    off = 8 + 32
    for _ in range(3):
        off += 8 + 4 * int.from_bytes(raw[off:off + 8], "little")
    assert int.from_bytes(raw[off:off + 8], "little") == 1024 and int.from_bytes(raw[off + 8:off + 16], "little") == prm.Q
    for damage in ("magic", "commit", "short", "ring degree", "modulus"):
        bad = bytearray(raw)
        if damage == "magic":
            bad[0] ^= 1
        elif damage == "ring degree":
            bad[off:off + 8] = (2048).to_bytes(8, "little")          # 5 N + 1 public inputs no longer fit the usable rows
        elif damage == "modulus":
            bad[off + 8:off + 16] = (1 << 63).to_bytes(8, "little")  # outside the range the witness kernels divide by
        elif damage == "commit":
            bad[len(raw) - (163 + 199) * 8192 * 32 - 8 - 32 - 64] ^= 1     # inside the last sigma commitment
        else:
            bad = bad[: len(bad) // 2]
        p2 = str(tmp_path / ("bad_%s.pk" % damage.replace(" ", "_")))
        open(p2, "wb").write(bytes(bad))
        with pytest.raises(zk.ZkfheError):
            zk.BfvProvingKey.load(ctx, srs, p2, 1024)
    srs2 = zk.Srs(ctx, 12)
    with pytest.raises(zk.ZkfheError):
        zk.BfvProvingKey.load(ctx, srs2, path, 1024)
    srs2.destroy()
    # a key is bound to the SRS it was generated with: the same k from another seed is refused (two of its commitments are
    # recomputed against the SRS it is loaded with), not accepted to produce proofs that never verify
    srs3 = zk.Srs(ctx, 13, seed=b"another-ceremony")
    with pytest.raises(zk.ZkfheError, match="different SRS"):
        zk.BfvProvingKey.load(ctx, srs3, path, 1024)
    srs3.destroy()
    # header fields that size buffers: unusable_rows (u32 at byte 28) and lookup_bits (byte 32) out of range
    for off_b, val in ((8 + 5 * 4, 2), (8 + 5 * 4, 1 << 20), (8 + 6 * 4, 0), (8 + 6 * 4, 31)):
        bad = bytearray(raw)
        bad[off_b:off_b + 4] = int(val).to_bytes(4, "little")
        p3 = str(tmp_path / "bad_hdr.pk")
        open(p3, "wb").write(bytes(bad))
        with pytest.raises(zk.ZkfheError, match="header"):
            zk.BfvProvingKey.load(ctx, srs, p3, 1024)
    srs.destroy()


def test_external_srs_points(ctx):
    """zkfhe_srs_from_points: an SRS computed elsewhere (here: by the oracle, same secret as the seeded setup) gives the same
    key and the same proof bytes as zkfhe_srs_create."""
    import zk_fhe_amd as zk
    prm = C.BfvParams(N=8)
    inp = synth_input(8, prm.Q, prm.T, prm.B, 1)
    circ = H.BfvCircuit(inp, prm)
    hcfg = H.auto_config(9, 9, circ)
    srs_o = H.make_srs(9)
    zcfg = zk.BfvConfig(9, hcfg.n_gate0, hcfg.n_gate1, hcfg.n_lookup, hcfg.n_rlc, 9)
    out = []
    for srs in (zk.Srs(ctx, 9), zk.Srs.from_points(ctx, 9, srs_o["g"], srs_o["g_lagrange"])):
        pk = zk.BfvProvingKey(ctx, srs, json.dumps(inp), (8, prm.Q, prm.T, prm.B), zcfg)
        out.append((pk.info(), pk.prove(json.dumps(inp), b"ext")[0]))
        pk.destroy()
        srs.destroy()
    assert out[0] == out[1]


def test_concurrent_proofs_on_two_streams(ctx):
    """Two contexts (streams + workspaces) of the same GPU prove against one key at the same time: same bytes as alone."""
    import threading
    import zk_fhe_amd as zk
    prm = C.BfvParams(N=8)
    inputs = [synth_input(8, prm.Q, prm.T, prm.B, s) for s in (1, 2, 3, 4)]
    circ = H.BfvCircuit(inputs[0], prm)
    hcfg = H.auto_config(9, 9, circ)
    srs = zk.Srs(ctx, 9)
    pk = zk.BfvProvingKey(ctx, srs, json.dumps(inputs[0]), (8, prm.Q, prm.T, prm.B), zk.BfvConfig(9, hcfg.n_gate0, hcfg.n_gate1, hcfg.n_lookup, hcfg.n_rlc, 9))
    alone = [pk.prove(json.dumps(i), b"s%d" % k)[0] for k, i in enumerate(inputs)]
    ctx2 = zk.Context(0)
    got = [None] * 4

    def work(c, ks):
        for _ in range(3):
            for k in ks:
                got[k] = pk.prove(json.dumps(inputs[k]), b"s%d" % k, ctx=c)[0]
    t1 = threading.Thread(target=work, args=(ctx, (0, 1)))
    t2 = threading.Thread(target=work, args=(ctx2, (2, 3)))
    t1.start(), t2.start()
    t1.join(), t2.join()
    assert got == alone
    pk.destroy()
    srs.destroy()
    ctx2.close()


def test_twelve_concurrent_k13_proofs_match_sequential(ctx):
    """The bench configuration: 12 contexts prove different k = 13 inputs against one key at the same time (zk_fhe_amd.batch
    .run_concurrent); every proof must equal the one the same input and seed give alone, and the C++ verifier accepts a sample."""
    import zk_fhe_amd as zk
    import zk_fhe_amd.batch as batch
    from zk_fhe_amd import inputs as gen
    cfgj = json.load(open(os.path.join(G, "bfv_config.json")))
    prm = C.BfvParams()
    srs = zk.Srs(ctx, 13)
    pk = zk.BfvProvingKey(ctx, srs, json.dumps(gen.empty(1024)), (1024, prm.Q, prm.T, prm.B), zk.BfvConfig.from_pinning(cfgj), replay=True)
    texts = [json.dumps(gen.generate(1024, prm.Q, prm.T, prm.B, seed=100 + i)) for i in range(6)]
    jobs = list(range(36))
    alone = {j: pk.prove(texts[j % 6], b"c%d" % j)[0] for j in jobs[:12]}
    ctxs = [ctx] + [zk.Context(0) for _ in range(11)]
    got = batch.run_concurrent(jobs, ctxs, lambda c, j: pk.prove(texts[j % 6], b"c%d" % j, ctx=c))
    for j in jobs[:12]:
        assert got[j][0] == alone[j], "proof %d differs when proved concurrently" % j
    # the same again with the transcripts' long runs (public inputs, commitments, evaluations) going through the shared eight-lane
    # Poseidon service (host/poseidon_x8.cpp; zkfhe_host_hash_mode): the same bytes
    # ... and with the admission gate of the heavy middle of a proof closed to three at a time (zkfhe_prover_gate): scheduling only
    if zk.poseidon_hash_many([[1, 2]], mode=1) is not None:
        assert zk.host_hash_mode("shared") == "shared"
        assert zk.prover_gate(3) == 0 and zk.prover_gate() == 3
        try:
            shared = batch.run_concurrent(jobs[:24], ctxs, lambda c, j: pk.prove(texts[j % 6], b"c%d" % j, ctx=c))
        finally:
            zk.host_hash_mode("latency")
            assert zk.prover_gate(0) == 3
        for j in jobs[:24]:
            assert shared[j][0] == got[j][0], "proof %d differs with the shared hash service" % j
    # jobs 12.. repeat the inputs with other seeds: different bytes, same instances, all valid
    vk = pk.export_vk()
    for j in (12, 23, 35):
        assert got[j][1] == got[j % 6][1]
        assert got[j][0] != got[j % 6][0]
        ok, why = zk.bfv_verify(vk, got[j][1], got[j][0])
        assert ok, why
    for c in ctxs[1:]:
        c.close()
    pk.destroy()
    srs.destroy()


def test_k14_proof_bytes_match_oracle(ctx):
    """A circuit with more rows than one NTT tile (k = 14 > 13): exercises the long-row NTT / coset paths in the prover."""
    import zk_fhe_amd as zk
    prm = C.BfvParams(N=16)
    inp = synth_input(16, prm.Q, prm.T, prm.B, 5)
    circ = H.BfvCircuit(inp, prm)
    hcfg = H.auto_config(14, 109, circ)
    srs_o = H.make_srs(14)
    pk_o, _ = H.keygen_circuit(hcfg, circ, srs_o)
    proof_o, inst_o = H.prove(hcfg, pk_o, srs_o, circ, b"k14")
    assert H.verify(H.VerifyingKey(pk_o), srs_o, inst_o, proof_o)
    srs = zk.Srs(ctx, 14)
    pk = zk.BfvProvingKey(ctx, srs, json.dumps(inp), (16, prm.Q, prm.T, prm.B),
                          zk.BfvConfig(14, hcfg.n_gate0, hcfg.n_gate1, hcfg.n_lookup, hcfg.n_rlc, 109))
    assert pk.info()["vk_digest"] == pk_o.vk_digest
    proof, inst, _ = pk.prove(json.dumps(inp), b"k14")
    assert first_diff(proof, proof_o) is None, "first differing 32-byte item: %s" % first_diff(proof, proof_o)
    pk.destroy()
    srs.destroy()


def native_cpu_proof(pk, hcfg, prm, text, seed, tmp_dir):
    """The same proof by the native CPU prover (oracle/cpu_prover.cpp: the oracle's 64-bit-limb arithmetic, its own Pippenger,
    NTT, quotient and SHPLONK loops), fed with the fixed / sigma columns of the GPU key (read back from the key file, whose
    last section holds them) and the oracle's SRS for the same seed.  Parity of the PROVER at sizes the Python oracle prover
    cannot reach in reasonable time; the keygen itself is compared with the oracle's at k = 13 / 14."""
    import types
    import numpy as np
    from oracle import binding as orc
    from oracle import cpu_prover as CP
    n = hcfg.n
    path = os.path.join(tmp_dir, "key.pk")
    pk.save(path)
    tail = (hcfg.n_fixed + hcfg.n_perm) * n * 32
    with open(path, "rb") as f:
        f.seek(-tail, os.SEEK_END)
        cols = np.frombuffer(f.read(tail), dtype=np.uint64).reshape(hcfg.n_fixed + hcfg.n_perm, n, 4)
    os.remove(path)
    fixed_l, sigma_l = cols[: hcfg.n_fixed], cols[hcfg.n_fixed:]
    lag = np.zeros((3, n, 4), dtype=np.uint64)
    one = H.M(1)
    lag[0, 0] = one
    lag[1, hcfg.u] = one
    lag[2, : hcfg.u] = one
    info = pk.info()
    key = types.SimpleNamespace(fixed_lagrange=fixed_l, sigma_lagrange=sigma_l, fixed_coeff=orc.ntt(fixed_l, hcfg.k, True),
                                sigma_coeff=orc.ntt(sigma_l, hcfg.k, True), l_coeff=orc.ntt(lag, hcfg.k, True),
                                vk_digest=info["vk_digest"], break_points=info["break_points"])
    cp = CP.CpuProver(hcfg, key, H.make_srs(hcfg.k), prm)
    CP.set_threads(CP.usable_cpus())
    proof = cp.prove(text, seed)
    print("native CPU prover, k = %d: %s ms" % (hcfg.k, {k2: round(v) for k2, v in cp.phase_ms.items()}))
    cp.close()
    return proof


def _prove_and_verify_large(ctx, N, k, tag, tmp_dir=None):
    """Large configurations: the Python oracle prover takes minutes to hours here, so it ran ONCE on the CPU and its proof is a
    committed digest (tests/golden/large_proofs.json, round 5): the GPU proof must have the oracle's bytes.  Besides: the oracle
    VERIFIER (pairing check) accepts it against the commitments of the GPU keygen, a tampered proof is refused, and (tmp_dir given)
    the proof equals the native CPU prover's byte for byte."""
    import time
    import zk_fhe_amd as zk
    from tests.large_inputs import B, Q60 as Q, T, large_input
    t_ = [time.time()]

    def lap(what):   # where the minutes of the large configurations go (shown with pytest -s)
        t_.append(time.time())
        print("%s: %-28s %6.1f s" % (tag, what, t_[-1] - t_[-2]))
    inp = large_input(N)
    text = json.dumps(inp)
    # column counts: place the circuit with generous limits and count the break points (halo2-base auto-configuration)
    probe = zk.bfv_build_tables(text, (N, Q, T, B), zk.BfvConfig(k, 8, 400, 120, 16, 109), 1, keygen_mode=False)
    n0, n1, nr = (len(probe["break_points"][k]) + 1 for k in ("gate0", "gate1", "rlc"))
    nl = -(-probe["lookups"] // ((1 << k) - 109))
    print(tag, "columns: gate", n0, n1, "lookup", nl, "rlc", nr, "cells", probe["cells"])
    zcfg = zk.BfvConfig(k, n0, n1, nl, nr, 109)
    lap("input + column counts")
    srs = zk.Srs(ctx, k)
    lap("SRS")
    pk = zk.BfvProvingKey(ctx, srs, text, (N, Q, T, B), zcfg)
    info = pk.info()
    lap("keygen")
    proof, inst, tm = pk.prove(text, tag.encode())
    lap("prove")
    print(tag, "prove timings ms [witness, commit, quotient, open, total]:", tm, "proof bytes", len(proof))
    assert len(inst) == 4 * N + N + 1
    # the ORACLE prover's proof of the same input and seed (Python, minutes to hours at this size: made once on the CPU by
    # tests/golden/gen_large_proof_digests.py, committed as digests): column counts, verifying-key digest, public inputs and proof bytes
    gold = json.load(open(os.path.join(HERE, "golden", "large_proofs.json"))).get("k%d" % k)
    if gold is not None:
        import hashlib
        assert (gold["N"], gold["seed"], gold["columns"]) == (N, tag, [n0, n1, nl, nr])
        assert "%064x" % info["vk_digest"] == gold["vk_digest"], "verifying key differs from the oracle keygen's"
        assert hashlib.sha256(b"".join(int(v).to_bytes(32, "little") for v in inst)).hexdigest() == gold["instances_sha256"]
        assert len(proof) == gold["proof_len"] and proof[:64].hex() == gold["proof_head_hex"] and proof[-64:].hex() == gold["proof_tail_hex"]
        assert hashlib.sha256(proof).hexdigest() == gold["proof_sha256"], "proof bytes differ from the oracle prover's"
    hcfg = H.Config(k, n0, n1, nl, nr, 109)
    vk = H.RawVerifyingKey(hcfg, info["fixed_commit"], info["sigma_commit"], info["vk_digest"])
    srs_v = H.srs_verifier_half(k)
    lap("verifier-side SRS (oracle)")
    assert H.verify(vk, srs_v, inst, proof)
    lap("oracle verifier")
    vkb = pk.export_vk()
    ok, why = zk.bfv_verify(vkb, inst, proof)
    assert ok, why
    lap("C++ verifier")
    if tmp_dir is not None:
        import types
        proof_c = native_cpu_proof(pk, hcfg, types.SimpleNamespace(N=N, Q=Q, T=T, B=B), text, tag.encode(), str(tmp_dir))
        assert first_diff(proof, proof_c) is None, "first differing 32-byte item: %s" % first_diff(proof, proof_c)
        lap("native CPU prover")
    # negative coverage at this size: a flipped bit in a commitment, an evaluation and both opening points, and a changed
    # public input -- each rejected by the oracle verifier and by the C++ verifier
    for pos in (5, len(proof) // 2, len(proof) - 40, len(proof) - 1):
        bad = bytearray(proof)
        bad[pos] ^= 1
        assert not H.verify(vk, srs_v, inst, bytes(bad)), "oracle verifier accepts a proof tampered at byte %d" % pos
        assert not zk.bfv_verify(vkb, inst, bytes(bad))[0], "C++ verifier accepts a proof tampered at byte %d" % pos
    inst2 = list(inst)
    inst2[N + 2] = (inst2[N + 2] + 1) % H.R
    assert not zk.bfv_verify(vkb, inst2, proof)[0]
    if k <= 16:   # the pure-Python sponge over other public inputs starts from scratch: 5.6 s at k = 16, 22 s at k = 19 -- the oracle's
        assert not H.verify(vk, srs_v, inst2, proof)   # answer on a changed input is checked at every size up to 2^16
    lap("tampered proofs, both verifiers")
    pk.destroy()
    srs.destroy()
    return (n0, n1, nl, nr), probe["cells"]


def test_config4_k16_n4096_60bit_modulus(ctx, tmp_path):
    """BASELINE config 4: N = 4096, 60-bit Q (witness values up to 132 bits), k = 16: accepted by both verifiers and
    byte-identical to the native CPU prover's proof."""
    cols, cells = _prove_and_verify_large(ctx, 4096, 16, "config4", tmp_path)
    assert cols == (2, 124, 34, 3) and cells[1] == 8073420     # the shape SURVEY.md section 8d predicts


def test_config5_k19_n16384(ctx, tmp_path):
    """BASELINE config 5: N = 16384, k = 19 (n = 524288 rows): MSM-dominated, long-row NTTs everywhere.  The proof must have the bytes
    of the Python oracle prover's (tests/golden/large_proofs.json: a prover that shares no line with the product) and pass both
    verifiers.  Round 6: the native CPU prover's comparison stays at k = 16 and k = 13 -- at this size it costs a minute (its own SRS
    from the oracle: 2^20 scalar multiplications on the host) and adds nothing to byte equality with the independent oracle."""
    cols, cells = _prove_and_verify_large(ctx, 16384, 19, "config5", None)
    assert cols[1] <= 64


@pytest.mark.parametrize("transcript", ["poseidon", "blake2b"])
def test_prefix_cache_hit_and_miss_give_the_same_bytes(ctx, transcript):
    """The per-public-key transcript cache (host/prefix_cache.hpp, zkfhe_bfv_pk_prefix_cache): a proof that restores the state
    behind `vk digest | pk0 | pk1` (hit), one that computes and stores it (miss) and one made with the cache off are the same
    bytes; two public keys interleaved, several encryptions each, sequentially and with four proofs in flight; the counters
    count.  Toy circuit (k = 9, N = 8: a 16-value key prefix of 41 public inputs), both transcripts."""
    import zk_fhe_amd as zk
    import zk_fhe_amd.batch as batch
    from zk_fhe_amd import inputs as gen
    prm = C.BfvParams(N=8)
    par = (8, prm.Q, prm.T, prm.B)
    texts = {(key, s): json.dumps(gen.generate(8, prm.Q, prm.T, prm.B, seed=s, key_seed=key)) for key in (100, 200) for s in (1, 2, 3)}
    order = [(100, 1), (200, 1), (100, 2), (200, 2), (100, 3), (200, 3), (100, 1)]
    cfg = zk.bfv_auto_config(texts[(100, 1)], par, 9, unusable_rows=9, transcript=transcript)
    srs = zk.Srs(ctx, 9)
    pk = zk.BfvProvingKey(ctx, srs, texts[(100, 1)], par, cfg)
    assert pk.prefix_cache() == {"hits": 0, "misses": 0, "entries": 0}
    pk.prefix_cache(0)
    plain = [pk.prove(texts[o], b"pc-%d" % i)[0] for i, o in enumerate(order)]
    assert pk.prefix_cache() == {"hits": 0, "misses": 0, "entries": 0}            # off: nothing looked up, nothing stored
    pk.prefix_cache(8)
    cached = [pk.prove(texts[o], b"pc-%d" % i)[0] for i, o in enumerate(order)]
    assert cached == plain
    assert pk.prefix_cache() == {"hits": 5, "misses": 2, "entries": 2}            # one miss per public key
    # four proofs in flight on their own contexts, all hits now
    ctxs = [zk.Context(0) for _ in range(4)]
    conc = batch.run_concurrent(list(range(len(order))), ctxs, lambda c, i: pk.prove(texts[order[i]], b"pc-%d" % i, ctx=c)[0])
    assert conc == plain and pk.prefix_cache()["hits"] == 12
    # one remembered key: the two keys evict each other, still the same bytes
    pk.prefix_cache(1)
    assert pk.prefix_cache()["entries"] == 1
    again = [pk.prove(texts[o], b"pc-%d" % i)[0] for i, o in enumerate(order)]
    assert again == plain
    st = pk.prefix_cache()
    # the keys alternate: every proof misses, except the first when its key is the entry that survived the shrink
    assert st["entries"] == 1 and st["hits"] in (12, 13) and st["hits"] + st["misses"] == 12 + 2 + len(order)
    for c in ctxs:
        c.close()
    pk.destroy()
    srs.destroy()


def test_prefix_cache_hit_matches_the_oracle_proof_k13(ctx):
    """bfv.in at k = 13 (2 048 of the 5 121 public inputs are the key): the first proof misses and stores, the second hits, a
    third with the cache off recomputes -- all three are the ORACLE prover's bytes for that seed."""
    import zk_fhe_amd as zk
    o = oracle_k13()
    srs = zk.Srs(ctx, 13)
    pk = zk.BfvProvingKey(ctx, srs, o["text_empty"], (1024, o["prm"].Q, o["prm"].T, o["prm"].B), zk.BfvConfig.from_pinning(o["cfgj"]))
    for want in ({"hits": 0, "misses": 1, "entries": 1}, {"hits": 1, "misses": 1, "entries": 1}):
        proof, inst, _ = pk.prove(o["text"], b"seed-1")
        assert proof == o["proof_o"] and inst == o["inst_o"]
        assert pk.prefix_cache() == want
    pk.prefix_cache(0)
    assert pk.prove(o["text"], b"seed-1")[0] == o["proof_o"]
    pk.destroy()
    srs.destroy()


def test_second_srs_on_a_full_device_gets_narrower_tables_same_bytes(ctx, monkeypatch):
    """Behaviour under memory pressure (README "Table budget"): with the service profile (ZKFHE_TABLE_GB=160) one k = 13 SRS takes
    189 GB of the 288; a SECOND one made while it is alive cannot have that -- the library leaves a reserve for keys and workspaces,
    narrows the second SRS's tables until they fit, says so (zkfhe_srs_table_info: narrowed) and proves the same bytes with it."""
    import zk_fhe_amd as zk
    monkeypatch.delenv("ZKFHE_TABLE_BITS", raising=False)
    monkeypatch.setenv("ZKFHE_TABLE_GB", "160")
    o = oracle_k13()
    a = zk.Srs(ctx, 13)
    ia = a.table_info()
    if ia["bits"][1] < 15:
        a.destroy()
        pytest.skip("the device is not empty enough for the 189 GB profile: %s" % ia)
    b = zk.Srs(ctx, 13)
    ib = b.table_info()
    print("first SRS %s, second SRS %s" % (ia, ib))
    assert not ia["narrowed"] and ib["narrowed"] and ib["bits"][1] < 15 and ib["gb"] < ia["gb"]
    cfg = zk.BfvConfig.from_pinning(o["cfgj"])
    for srs in (a, b):
        pk = zk.BfvProvingKey(ctx, srs, o["text_empty"], (1024, o["prm"].Q, o["prm"].T, o["prm"].B), cfg)
        assert pk.prove(o["text"], b"seed-1")[0] == o["proof_o"]
        pk.destroy()
    b.destroy()
    a.destroy()


@pytest.mark.parametrize("transcript", ["poseidon", "blake2b"])
def test_announced_proofs_have_the_same_bytes(ctx, transcript):
    """zkfhe_bfv_pk_prehash (host/prefix_cache.hpp PreHash): the public inputs of a LATER proof absorbed ahead of time on a helper
    thread.  A proof that starts from the parked state has the bytes of one that hashes for itself; an announcement serves exactly
    one proof; announcing input X does not touch a proof of input Y; two announcements of one input serve two proofs; with the
    per-key cache on or off; four proofs in flight, each announced; an input that does not parse serves nobody; the queue is bounded.
    Toy circuit, both transcripts."""
    import zk_fhe_amd as zk
    import zk_fhe_amd.batch as batch
    from zk_fhe_amd import inputs as gen
    prm = C.BfvParams(N=8)
    par = (8, prm.Q, prm.T, prm.B)
    texts = [json.dumps(gen.generate(8, prm.Q, prm.T, prm.B, seed=s, key_seed=100 + s % 2)) for s in range(6)]
    cfg = zk.bfv_auto_config(texts[0], par, 9, unusable_rows=9, transcript=transcript)
    srs = zk.Srs(ctx, 9)
    pk = zk.BfvProvingKey(ctx, srs, texts[0], par, cfg)
    plain = [pk.prove(t, b"ann-%d" % i)[0] for i, t in enumerate(texts)]
    assert pk.prehash() == {"started": 0, "taken": 0, "pending": 0}
    for cache in (8, 0):
        pk.prefix_cache(cache)
        base = pk.prehash()["started"]
        assert pk.prehash(texts[1])["pending"] == 1
        assert pk.prove(texts[0], b"ann-0")[0] == plain[0] and pk.prehash()["pending"] == 1      # another input: not taken
        assert pk.prove(texts[1], b"ann-1")[0] == plain[1] and pk.prehash()["pending"] == 0      # taken
        assert pk.prove(texts[1], b"ann-1")[0] == plain[1]                                       # one-shot: this one hashed for itself
        pk.prehash(texts[2]), pk.prehash(texts[2])
        assert pk.prove(texts[2], b"ann-2")[0] == plain[2] and pk.prove(texts[2], b"ann-2")[0] == plain[2]
        st = pk.prehash()
        assert st["started"] == base + 3 and st["pending"] == 0 and st["taken"] == st["started"]
    pk.prefix_cache(8)
    ctxs = [zk.Context(0) for _ in range(4)]
    for t in texts:
        pk.prehash(t)

    def job(c, i):
        return pk.prove(texts[i], b"ann-%d" % i, ctx=c)[0]
    assert batch.run_concurrent(list(range(len(texts))), ctxs, job) == plain
    st = pk.prehash()
    assert st["pending"] == 0 and st["taken"] == st["started"]
    # an input that does not parse is announced like any other (the call only copies the text) and serves nobody; its entry goes when the
    # next announcement finds it failed; seventeen pending announcements are one too many
    import time
    pk.prehash('{"pk0": ["1"]}')
    time.sleep(0.5)
    for _ in range(16):
        pk.prehash(texts[0])
    assert pk.prehash()["pending"] == 16
    with pytest.raises(zk.ZkfheError):
        pk.prehash(texts[0])
    assert pk.prove(texts[0], b"ann-0")[0] == plain[0] and pk.prehash()["pending"] == 15
    for c in ctxs:
        c.close()
    pk.destroy()
    srs.destroy()


def test_announced_proof_matches_the_oracle_proof_k13(ctx):
    """bfv.in at k = 13 announced ahead of time: the proof that picks the parked state up (5 121 public inputs absorbed on the helper
    thread, through the per-key prefix) is the ORACLE prover's bytes, and so is the next one, which hashes for itself."""
    import zk_fhe_amd as zk
    o = oracle_k13()
    srs = zk.Srs(ctx, 13)
    pk = zk.BfvProvingKey(ctx, srs, o["text_empty"], (1024, o["prm"].Q, o["prm"].T, o["prm"].B), zk.BfvConfig.from_pinning(o["cfgj"]))
    pk.prehash(o["text"])
    proof, inst, _ = pk.prove(o["text"], b"seed-1")
    assert proof == o["proof_o"] and inst == o["inst_o"]
    assert pk.prehash() == {"started": 1, "taken": 1, "pending": 0}
    marks = ctx.last_proof_marks()
    assert 0 < marks[0] <= marks[1] <= marks[2]
    assert pk.prove(o["text"], b"seed-1")[0] == o["proof_o"]
    pk.destroy()
    srs.destroy()
