"""Host (C++) witness generator behind the C ABI vs the oracle restatement and the reference's pinned layout.
CPU only: zkfhe_bfv_build_tables does not touch the GPU."""
import json
import os

import numpy as np
import pytest

import zk_fhe_amd as zk
from oracle import binding as orc
from oracle import circuit_ref as C
from oracle import halo2_ref as H

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden", "bfv")
PRM = (1024, 536870909, 7, 19)
GAMMA = 0x0123456789ABCDEF0FEDCBA9876543210123456789ABCDEF0FEDCBA987654321 % H.R


def ints(arr):
    return orc.arr_to_ints(np.ascontiguousarray(arr).reshape(-1, 4))


@pytest.fixture(scope="module")
def pinned():
    cfgj = json.load(open(os.path.join(G, "bfv_config.json")))
    return cfgj, zk.BfvConfig.from_pinning(cfgj), H.Config.from_pinning(cfgj)


@pytest.mark.parametrize("fname", ["bfv.in", "bfv_empty.in"])
def test_tables_match_oracle_and_pinned_layout(pinned, fname):
    cfgj, zcfg, hcfg = pinned
    text = open(os.path.join(G, fname)).read()
    # keygen stage: break points are COMPUTED and must equal the reference's configs/bfv.json
    zcfg_nobp = zk.BfvConfig(zcfg.k, zcfg.n_gate0, zcfg.n_gate1, zcfg.n_lookup, zcfg.n_rlc, zcfg.unusable_rows, zcfg.lookup_bits)
    t = zk.bfv_build_tables(text, PRM, zcfg_nobp, GAMMA, keygen_mode=True)
    assert t["break_points"]["gate0"] == cfgj["break_points"]["gate"][0]
    assert t["break_points"]["gate1"] == cfgj["break_points"]["gate"][1]
    assert t["break_points"]["rlc"] == cfgj["break_points"]["rlc"]
    assert t["cells"] == (23558, 1231992, 32764) and t["lookups"] == 286756
    assert t["instance"].shape[0] == 5121
    # oracle on the same input and challenge
    prm = C.BfvParams()
    inp = json.loads(text)
    ctx0, pub, st = C.bfv_phase0(inp, prm)
    ctx_gate, ctx_rlc = C.bfv_phase1(st, prm, GAMMA)
    A = H.assign(hcfg, ctx0, ctx_gate, ctx_rlc, pub)
    got_adv = ints(t["advice"])
    want_adv = [v for col in A.advice for v in col]
    assert got_adv == want_adv
    assert ints(t["fixed"]) == [v for col in A.fixed for v in col]
    assert ints(t["instance"]) == A.instance
    n = hcfg.n
    want_cp = sorted((min(a[0] * n + a[1], b[0] * n + b[1]), max(a[0] * n + a[1], b[0] * n + b[1])) for a, b in A.copies)
    got_cp = sorted((int(min(a, b)), int(max(a, b))) for a, b in t["copies"])
    assert got_cp == want_cp
    # prover stage: replaying the pinned break points, values only
    t2 = zk.bfv_build_tables(text, PRM, zcfg, GAMMA, keygen_mode=False, replay=True)
    assert np.array_equal(t2["advice"], t["advice"])
    assert t2["copies"].shape[0] == 0


def test_bad_inputs_fail_with_status():
    cfgj = json.load(open(os.path.join(G, "bfv_config.json")))
    zcfg = zk.BfvConfig.from_pinning(cfgj)
    inp = json.load(open(os.path.join(G, "bfv.in")))
    bad = dict(inp)
    bad["pk0"] = inp["pk0"][:-1]  # wrong degree -> assert_eq!(deg, N-1) of examples/bfv.rs:82
    with pytest.raises(zk.ZkfheError):
        zk.bfv_build_tables(json.dumps(bad), PRM, zcfg, 1, True)
    bad = dict(inp)
    bad["u"] = ["536870910"] + inp["u"][1:]  # coeff > modulus -> assert of src/poly.rs:28
    with pytest.raises(zk.ZkfheError):
        zk.bfv_build_tables(json.dumps(bad), PRM, zcfg, 1, True)
    with pytest.raises(zk.ZkfheError):
        zk.bfv_build_tables("{not json", PRM, zcfg, 1, True)
    # the form errors of the input file, each named
    text = json.dumps(inp)
    for broken, what in ((text.replace('"cyclo"', '"cyclone"'), "missing field cyclo"),
                         (text[: text.index("]")], "expected ']'"),
                         (text.replace('"pk1": [', '"pk1": '), r"expected '\['"),
                         (text.replace('"m": ["', '"m": [', 1), "expected string"),
                         (json.dumps(dict(inp, e0=[])), "at least one coefficient")):
        with pytest.raises(zk.ZkfheError, match=what):
            zk.bfv_build_tables(broken, PRM, zcfg, 1, True)
    # fields the circuit does not read are ignored; a repeated field counts once (the last occurrence)
    extra = json.dumps(dict(inp, note=["1", "2"]))
    assert zk.bfv_build_tables(extra, PRM, zcfg, 1, False)["cells"] == zk.bfv_build_tables(text, PRM, zcfg, 1, False)["cells"]


def test_long_inputs_parse_on_threads_like_short_ones():
    """N = 4096 inputs (above the 256 KB mark) take the threaded parse and the threaded decimal conversions: same tables as the
    same numbers re-serialised with different white space, and the form errors still surface from the worker threads."""
    from zk_fhe_amd import inputs as gen
    N, Q = 4096, (1 << 60) - 93
    inp = gen.generate(N, Q, 7, 19, seed=5)
    compact = json.dumps(inp, separators=(",", ":"))
    spaced = json.dumps(inp, indent=1)
    assert len(compact) > (1 << 18)
    cfg = zk.BfvConfig(16, 8, 400, 120, 16, 109)
    a = zk.bfv_build_tables(compact, (N, Q, 7, 19), cfg, 3, keygen_mode=False)
    b = zk.bfv_build_tables(spaced, (N, Q, 7, 19), cfg, 3, keygen_mode=False)
    assert a["cells"] == b["cells"] and (a["instance"] == b["instance"]).all() and (a["advice"] == b["advice"]).all()
    with pytest.raises(zk.ZkfheError, match="expected string"):
        zk.bfv_build_tables(compact.replace('"c1":["', '"c1":[', 1), (N, Q, 7, 19), cfg, 3, keygen_mode=False)
    with pytest.raises(zk.ZkfheError, match="coeff <= modulus"):
        zk.bfv_build_tables(json.dumps(dict(inp, m=[str(Q + 1)] + inp["m"][1:]), separators=(",", ":")), (N, Q, 7, 19), cfg, 3, keygen_mode=False)


def test_auto_config_reproduces_the_pinned_column_counts():
    """zkfhe_bfv_auto_config (halo2-base auto-configuration, the first half of the reference's keygen) on bfv_empty.in at k = 13
    gives exactly the column counts of the reference's configs/bfv.json; k too small is an error, larger k needs fewer columns."""
    import json
    import os
    import zk_fhe_amd as zk
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bfv")
    cfgj = json.load(open(os.path.join(G, "bfv_config.json")))
    want = zk.BfvConfig.from_pinning(cfgj)
    text = open(os.path.join(G, "bfv_empty.in")).read()
    prm = (1024, 536870909, 7, 19)
    got = zk.bfv_auto_config(text, prm, 13)
    assert (got.n_gate0, got.n_gate1, got.n_lookup, got.n_rlc) == (want.n_gate0, want.n_gate1, want.n_lookup, want.n_rlc) == (3, 153, 36, 5)
    big = zk.bfv_auto_config(text, prm, 15)
    assert big.n_gate1 < got.n_gate1 and big.n_lookup < got.n_lookup
    import pytest
    with pytest.raises(zk.ZkfheError):
        zk.bfv_auto_config(text, prm, 6)


def test_mock_checks_every_row(pinned):
    """`mock` (README.md:18-22, MockProver::assert_satisfied): the reference's bfv.in satisfies every gate, lookup and copy
    constraint row by row; a single changed cell is caught as the gate / lookup / copy violation it causes."""
    cfgj, zcfg, hcfg = pinned
    text = open(os.path.join(G, "bfv.in")).read()
    zcfg_nobp = zk.BfvConfig(zcfg.k, zcfg.n_gate0, zcfg.n_gate1, zcfg.n_lookup, zcfg.n_rlc, zcfg.unusable_rows, zcfg.lookup_bits)
    fails, first = zk.bfv_mock(text, PRM, zcfg_nobp, GAMMA)
    assert fails == 0, first
    t = zk.bfv_build_tables(text, PRM, zcfg_nobp, GAMMA, keygen_mode=True)
    fixed = np.array(ints(t["fixed"]), dtype=object).reshape(t["fixed"].shape[0], -1)
    # a gate output cell (row 3 of the gate enabled at row 0 of the first phase-1 gate column)
    col = zcfg.n_gate0
    assert fixed[col][0] == 1
    old = ints(t["advice"][col][3:4])[0]
    fails, first = zk.bfv_mock(text, PRM, zcfg_nobp, GAMMA, pokes=[(col, 3, (old + 1) % H.R)])
    assert fails >= 1 and "gate" in first and "row 0" in first
    # a looked-up cell pushed out of the 8-bit table: lookup failure (and the copy constraint to its source cell)
    lcol = zcfg.n_gate0 + zcfg.n_gate1
    fails, first = zk.bfv_mock(text, PRM, zcfg_nobp, GAMMA, pokes=[(lcol, 5, 256)])
    assert fails == 2 and "lookup" in first
    # an RLC accumulator cell
    rcol = lcol + zcfg.n_lookup
    old = ints(t["advice"][rcol][2:3])[0]
    fails, first = zk.bfv_mock(text, PRM, zcfg_nobp, GAMMA, pokes=[(rcol, 2, (old + 5) % H.R)])
    assert fails >= 1 and "RLC" in first
    # a public input cell changed consistently nowhere else: the copy constraint to the instance column breaks
    fails, first = zk.bfv_mock(text, PRM, zcfg_nobp, GAMMA, pokes=[(0, 0, 12345)])
    assert fails >= 1
    # a wrong ciphertext coefficient: witness generation goes through, is_equal yields 0 and its copy to the constant 1 fails
    bad = json.loads(text)
    bad["c0"][2] = str((int(bad["c0"][2]) + 1) % PRM[1])
    fails, first = zk.bfv_mock(json.dumps(bad), PRM, zcfg_nobp, GAMMA)
    assert fails >= 1 and "copy constraint" in first
    # a coefficient above the modulus never reaches the table: Poly::from_string asserts (src/poly.rs:28)
    bad["c0"][2] = str(PRM[1] + 1)
    with pytest.raises(zk.ZkfheError):
        zk.bfv_mock(json.dumps(bad), PRM, zcfg_nobp, GAMMA)


def test_host_poly_mul_u32_is_the_integer_product():
    """Poly::mul (src/poly.rs:75-103) for short narrow polynomials runs an NTT convolution over 2^64 - 2^32 + 1 on the host
    (host/poly_ntt64.hpp, what the k = 13 prover uses for pk_i * u): every coefficient against Python integers -- random
    32-bit operands, the all-maximum case (the largest sums: n * (2^32 - 1)^2), the ternary shape of u (0 / 1 / Q - 1), n = 2
    and the largest n; operands that do not fit are refused."""
    import ctypes
    import numpy as np
    import zk_fhe_amd as zk
    lib = zk.load_library()
    lib.zkfhe_host_poly_mul_u32.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_size_t] + [ctypes.c_void_p] * 2
    rng = np.random.default_rng(5)
    Q = 536870909

    def run(a, b):
        n = len(a)
        A, B = np.array(a, dtype=np.uint64), np.array(b, dtype=np.uint64)
        lo, hi = np.zeros(2 * n - 1, dtype=np.uint64), np.zeros(2 * n - 1, dtype=np.uint64)
        rc = lib.zkfhe_host_poly_mul_u32(A.ctypes.data, B.ctypes.data, n, lo.ctypes.data, hi.ctypes.data)
        return rc, [int(lo[i]) | (int(hi[i]) << 64) for i in range(2 * n - 1)]

    def want(a, b):
        c = [0] * (2 * len(a) - 1)
        for i, x in enumerate(a):
            if x:
                for j, y in enumerate(b):
                    c[i + j] += x * y
        return c

    cases = [([3, 5], [7, 11])]
    for n in (8, 256, 1024, 2048):
        cases.append(([int(x) for x in rng.integers(0, 1 << 32, n)], [int(x) for x in rng.integers(0, 1 << 32, n)]))
    cases.append(([(1 << 32) - 1] * 2048, [(1 << 32) - 1] * 2048))
    cases.append(([int(x) for x in rng.integers(0, Q, 1024)], [[0, 1, Q - 1][int(t)] for t in rng.integers(0, 3, 1024)]))
    for a, b in cases:
        rc, got = run(a, b)
        assert rc == 0 and got == want(a, b)
    assert run([1 << 32, 1], [1, 1])[0] != 0          # a coefficient of 33 bits
    assert run([1, 2, 3], [1, 2, 3])[0] != 0          # not a power of two
    assert run([1] * 4096, [1] * 4096)[0] != 0        # too long


def test_machine_word_phase0_equals_the_restatement(monkeypatch):
    """host/bfv_phase0_fast.hpp (what a proof runs: u64 / u128 arithmetic, closed-form division by x^N + 1) against bfv_phase0
    (src/poly.rs + src/poly_chip.rs restated on BigInt): the same advice table, instances and cell counts -- on the reference's
    files, on seeded encryptions, and on edge inputs: coefficients equal to Q (src/poly.rs:28 allows `<=`), leading zeros, an
    all-zero u (the zero-dividend branch of divide_by_cyclo), a `+` sign (outside the fast grammar: falls back, same result).
    ZKFHE_PHASE0=generic forces the restatement."""
    from zk_fhe_amd import inputs as gen
    cfgj = json.load(open(os.path.join(G, "bfv_config.json")))
    zcfg = zk.BfvConfig.from_pinning(cfgj)
    N, Q, T, B = PRM
    cases = [open(os.path.join(G, "bfv.in")).read(), open(os.path.join(G, "bfv_empty.in")).read()]
    for seed in (1, 2):
        cases.append(json.dumps(gen.generate(N, Q, T, B, seed=seed)))
    inp = gen.generate(N, Q, T, B, seed=3)
    edge = dict(inp)
    edge["pk0"] = [str(Q)] + ["000" + v for v in inp["pk0"][1:]]          # == Q passes the reference's assert; leading zeros
    edge["pk1"] = [str(Q - 1)] * N
    cases.append(json.dumps(edge))
    cases.append(json.dumps(dict(inp, e1=["+" + inp["e1"][0]] + inp["e1"][1:])))   # not in the fast grammar
    zero_u = dict(inp, u=["0"] * N)
    cases.append(json.dumps(zero_u))

    def both(text):
        out = []
        for mode in ("fast", "generic"):
            if mode == "generic":
                monkeypatch.setenv("ZKFHE_PHASE0", "generic")
            else:
                monkeypatch.delenv("ZKFHE_PHASE0", raising=False)
            try:
                t = zk.bfv_build_tables(text, PRM, zcfg, GAMMA, keygen_mode=False, replay=True)
                out.append((t["cells"], t["instance"].tobytes(), t["advice"].tobytes()))
            except zk.ZkfheError as e:
                out.append(str(e))
        return out
    n_ok = 0
    for text in cases:
        a, b = both(text)
        assert a == b
        n_ok += not isinstance(a, str)
    assert n_ok >= 5
    # errors are worded by the restatement whichever path is asked for
    for bad in (dict(inp, m=[str(Q + 1)] + inp["m"][1:]), dict(inp, c0=inp["c0"][:-1])):
        a, b = both(json.dumps(bad))
        assert isinstance(a, str) and a == b
    # a cyclo of another shape is outside the closed forms: the long division of the restatement runs, whatever it gives
    a, b = both(json.dumps(dict(inp, cyclo=["2"] + inp["cyclo"][1:])))
    assert a == b


def test_machine_word_phase0_wide_products(tmp_path):
    """N = 4096 with the 60-bit modulus (BASELINE configs[3]): the products pk_i * u have 132-bit coefficients (three words) and come
    from the PolyMulBackend -- the GPU convolution in a proof, a schoolbook product in tests/native/phase0_check.cpp.  Cell stream,
    public inputs, every PolyChip's cells, offsets and max_bits must equal the BigInt restatement's."""
    import shutil
    import subprocess
    from zk_fhe_amd import inputs as gen
    cxx = "/opt/rocm/lib/llvm/bin/clang++" if os.path.exists("/opt/rocm/lib/llvm/bin/clang++") else shutil.which("clang++")
    if not cxx:
        pytest.skip("no clang++ (the host headers use clang's carry builtins)")
    N, Q = 4096, (1 << 60) - 93
    path = tmp_path / "in.json"
    path.write_text(json.dumps(gen.generate(N, Q, 7, 19, seed=5)))
    exe = str(tmp_path / "phase0_check")
    subprocess.run([cxx, "-O2", "-std=c++17", "-I", os.path.join(HERE, "..", "zk-fhe_amd", "host"), os.path.join(HERE, "native", "phase0_check.cpp"), "-o", exe, "-lpthread"], check=True)
    out = subprocess.run([exe, str(path), str(N), str(Q)], capture_output=True, text=True)
    assert out.returncode == 0 and "fast path taken 1" in out.stdout and "identical 1" in out.stdout, out.stdout + out.stderr
