"""GPU parity tests: every C-ABI entry point against the CPU oracle (oracle/), bit-exact.
Run on the MI355X box:  python -m pytest tests -m gpu -x -q
"""
import json
import os

import numpy as np
import pytest

from oracle import binding as orc
from oracle import pyref

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def ctx():
    import torch  # noqa: F401  (loads the ROCm runtime the extension links against first)
    import zk_fhe_amd as zk
    c = zk.Context(0)
    assert c.device_info()["arch"].startswith("gfx950")
    yield c
    c.close()


@pytest.fixture(scope="module")
def vec():
    with open(os.path.join(HERE, "golden", "bn254_vectors.json")) as f:
        return json.load(f)


def rand_fr(rng, n):
    return orc.ints_to_mont([int.from_bytes(rng.bytes(32), "little") % pyref.R for _ in range(n)])


def hx(lst):
    return [int(x, 16) for x in lst]


def test_field_golden(ctx, vec):
    v = vec["fr"]
    a = orc.ints_to_mont(hx(v["a"]))
    b = orc.ints_to_mont(hx(v["b"]))
    for op in ("add", "sub", "mul"):
        assert orc.mont_to_ints(ctx.fr_binop(op, a, b)) == hx(v[op]), op
    assert orc.mont_to_ints(ctx.fr_unop("batch_invert", a)) == hx(v["inv"])
    # to/from Montgomery
    canon = orc.ints_to_arr(hx(v["a"]))
    assert np.array_equal(ctx.fr_unop("to_mont", canon), a)
    assert np.array_equal(ctx.fr_unop("from_mont", a), canon)


def test_field_random_vs_oracle(ctx):
    rng = np.random.default_rng(11)
    n = 100_003
    a, b = rand_fr(rng, n), rand_fr(rng, n)
    a[5] = 0
    for op in ("add", "sub", "mul"):
        assert np.array_equal(ctx.fr_binop(op, a, b), orc.fe_binop(op, a, b)), op
    assert np.array_equal(ctx.fr_unop("batch_invert", a), orc.fr_batch_inv(a))
    s = rand_fr(rng, 1)
    assert np.array_equal(ctx.fr_unop("scale", a, s), orc.fe_binop("mul", a, np.repeat(s, n, axis=0)))
    x = a[:1000]
    want = x.copy()
    for _ in range(5):
        want = orc.fe_binop("mul", want, want)
    assert np.array_equal(ctx.fr_unop("sqr_chain", x, 5), want)


def test_radix29_product_on_device(ctx):
    """fq29.hip.hpp on the GPU (the field arithmetic inside the MSM kernels): k squarings in the radix-2^29 Montgomery form
    (R' = 2^261) against python integers: x -> x^(2^k) * R'^-(2^k - 1) mod q, for random, tiny and q - 1 inputs."""
    import ctypes
    rng = np.random.default_rng(29)
    Q, n, iters = pyref.Q, 4096, 7
    xs = [int.from_bytes(rng.bytes(32), "little") % Q for _ in range(n)]
    xs[:4] = [0, 1, Q - 1, 2]
    mask = (1 << 64) - 1
    raw = np.array([[(v >> (64 * j)) & mask for j in range(4)] for v in xs], dtype=np.uint64)
    d, o = ctx.to_device(raw), ctx.alloc(n * 32)
    ctx.lib.zkfhe_fq29_sqr_chain.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    ctx._check(ctx.lib.zkfhe_fq29_sqr_chain(ctx.h, d.at(0), o.at(0), n, iters))
    got = o.download(shape=(n, 4))
    rinv = pow(pow(2, 261, Q), -1, Q)
    for v, g in zip(xs, got):
        w = v
        for _ in range(iters):
            w = w * w * rinv % Q
        assert sum(int(g[j]) << (64 * j) for j in range(4)) == w
    d.free(), o.free()


def test_g1_golden(ctx, vec):
    g = vec["g1"]
    G = orc.points_to_arr([pyref.G1_GEN] * len(g["k"]))
    k = orc.ints_to_mont(hx(g["k"]))
    want = [None if (int(x, 16) == 0 and int(y, 16) == 0) else (int(x, 16), int(y, 16)) for x, y in g["kG"]]
    assert orc.arr_to_points(ctx.g1_mul(G, k)) == want
    pts = orc.points_to_arr(want)
    A = np.stack([pts[c["i"]] for c in g["add"]])
    B = np.stack([pts[c["j"]] if c["j"] >= 0 else np.zeros(8, dtype=np.uint64) for c in g["add"]])
    got = orc.arr_to_points(ctx.g1_add(A, B))
    for s, c in zip(got, g["add"]):
        assert (s or (0, 0)) == (int(c["sum"][0], 16), int(c["sum"][1], 16))


@pytest.mark.parametrize("log_n", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13])
def test_ntt_vs_oracle(ctx, log_n):
    rng = np.random.default_rng(100 + log_n)
    n_cols = 5 if log_n < 13 else 3
    a = rand_fr(rng, n_cols << log_n).reshape(n_cols, 1 << log_n, 4)
    fwd = ctx.ntt(a, log_n, False)
    assert np.array_equal(fwd, orc.ntt(a, log_n, False))
    assert np.array_equal(ctx.ntt(a, log_n, True), orc.ntt(a, log_n, True))
    assert np.array_equal(ctx.ntt(fwd, log_n, True), a)


@pytest.mark.parametrize("log_n,n_cols", [(5, 3), (12, 4), (13, 1), (13, 37), (14, 2), (16, 5), (19, 3)])
def test_ntt_out_of_place(ctx, log_n, n_cols):
    """zkfhe_ntt_batch_to: the input buffer is left untouched, the output equals the oracle.  At 2^13 this is the kernel's native
    form (two workgroups per column, placed in groups of eight columns: 37 columns leave the last group ragged).  2^16 and 2^19:
    the three / six stages above the tile as ONE four-step pass (ntt_dif8.hip: size-8 / size-64 transforms on constants, one
    streaming table product per element), tiles numbered position-major over an odd number of columns."""
    rng = np.random.default_rng(900 + log_n + n_cols)
    n = 1 << log_n
    a = rand_fr(rng, n_cols * n).reshape(n_cols, n, 4)
    for inv in (False, True):
        src, dst = ctx.to_device(a), ctx.alloc(n_cols * n * 32)
        ctx.ntt_to_dev(src, dst, n_cols, log_n, inverse=inv)
        got = dst.download(shape=(n_cols, n, 4))
        assert np.array_equal(src.download(shape=(n_cols, n, 4)), a)
        want = orc.ntt(a, log_n, inv)
        assert np.array_equal(got, want)
        src.free(), dst.free()


def test_ntt_golden(ctx, vec):
    for t in vec["ntt"]:
        a = orc.ints_to_mont(hx(t["in"]))
        assert orc.mont_to_ints(ctx.ntt(a[None], t["log_n"], False)[0]) == hx(t["out"])


@pytest.mark.parametrize("log_n", [14, 15, 16, 17, 18, 19, 20])
def test_ntt_large(ctx, log_n):
    """Rows longer than the 2^13 tile: one, two or three fused stages per pass over memory (2^19: two passes of three, 2^20: three passes)."""
    rng = np.random.default_rng(log_n)
    a = rand_fr(rng, 2 << log_n).reshape(2, 1 << log_n, 4)
    assert np.array_equal(ctx.ntt(a, log_n, False), orc.ntt(a, log_n, False))
    assert np.array_equal(ctx.ntt(a, log_n, True), orc.ntt(a, log_n, True))


def test_coset_extension_of_long_rows(ctx):
    """2^19 rows on the coset of the extended domain (2 extension bits): the coset powers are multiplied in by the first pass as it
    loads."""
    rng = np.random.default_rng(1919)
    log_n, lef = 19, 2
    n, E = 1 << log_n, 1 << lef
    a = rand_fr(rng, n).reshape(1, n, 4)
    g = orc.ints_to_mont([pyref.FR_GEN])[0]
    got = ctx.coset_ntt(a, log_n, lef, g)
    nat = orc.coset_ntt(a[0], log_n + lef, g)
    assert np.array_equal(got[0], nat.reshape(n, E, 4).transpose(1, 0, 2).reshape(n * E, 4))


def test_ntt_k13_batch_properties(ctx):
    """BASELINE size (k=13, 197 witness columns): round trip + linearity, plus oracle on a column sample."""
    rng = np.random.default_rng(5)
    log_n, n_cols = 13, 197
    a = rand_fr(rng, n_cols << log_n).reshape(n_cols, 1 << log_n, 4)
    f = ctx.ntt(a, log_n, False)
    assert np.array_equal(ctx.ntt(f, log_n, True), a)
    for c in (0, 77, 196):
        assert np.array_equal(f[c], orc.ntt(a[c:c + 1], log_n, False)[0])
    s = orc.fe_binop("add", a[0], a[1])
    assert np.array_equal(ctx.ntt(s[None], log_n, False)[0], orc.fe_binop("add", f[0], f[1]))


@pytest.mark.parametrize("log_n,lef", [(4, 2), (7, 1), (10, 3), (13, 2), (14, 2), (15, 1), (16, 2), (16, 1)])
def test_coset_ntt(ctx, log_n, lef):
    rng = np.random.default_rng(log_n * 10 + lef)
    n, E = 1 << log_n, 1 << lef
    n_cols = 2
    a = rand_fr(rng, n_cols * n).reshape(n_cols, n, 4)
    g = orc.ints_to_mont([pyref.FR_GEN])[0]
    got = ctx.coset_ntt(a, log_n, lef, g)
    for c in range(n_cols):
        nat = orc.coset_ntt(a[c], log_n + lef, g)  # nat[k] = f(g w_ext^k)
        want = nat.reshape(n, E, 4).transpose(1, 0, 2).reshape(n * E, 4)  # [k1][k2] <- nat[k1 + E k2]
        assert np.array_equal(got[c], want)
    # inverse: random extended-domain values of a degree < n*E polynomial
    coeffs = rand_fr(rng, n_cols * n * E).reshape(n_cols, n * E, 4)
    ext = np.stack([orc.coset_ntt(coeffs[c], log_n + lef, g).reshape(n, E, 4).transpose(1, 0, 2).reshape(n * E, 4)
                    for c in range(n_cols)])
    back = ctx.coset_ntt(ext, log_n, lef, g, inverse=True)
    assert np.array_equal(back, coeffs)


def _bases(n, seed=1):
    return orc.g1_powers(orc.ints_to_mont([seed * 7919 + 3])[0], orc.ints_to_mont([0x1234567 + seed])[0], n)


def test_msm_golden(ctx, vec):
    import zk_fhe_amd as zk
    for m in vec["msm"]:
        bases = [None if (int(x, 16) == 0 and int(y, 16) == 0) else (int(x, 16), int(y, 16)) for x, y in m["bases"]]
        B = zk.Basis(ctx, orc.points_to_arr(bases))
        got = orc.arr_to_points(ctx.msm(B, orc.ints_to_mont(hx(m["scalars"]))[None]))[0]
        assert (got or (0, 0)) == (int(m["result"][0], 16), int(m["result"][1], 16))
        B.destroy()


@pytest.mark.parametrize("n,c", [(64, 0), (1000, 0), (1000, 4), (1000, 7), (4096, 0), (8192, 0), (8192, 10), (8192, 14), (8192, 16)])
def test_msm_vs_oracle(ctx, n, c):
    import zk_fhe_amd as zk
    rng = np.random.default_rng(n + c)
    bases = _bases(n, seed=n)
    bases[n // 3] = 0  # identity base
    n_cols = 4
    sc = [[int.from_bytes(rng.bytes(32), "little") % pyref.R for _ in range(n)] for _ in range(n_cols)]
    # column 1: witness-like small / negative-small values; column 2: heavily skewed (0/1/2^29-ish); column 3: zeros
    sc[1] = [int(rng.integers(0, 256)) if i % 3 else (pyref.R - int(rng.integers(1, 1000))) for i in range(n)]
    sc[2] = [[0, 1, 536870908][int(rng.integers(0, 3))] for _ in range(n)]
    sc[3] = [0] * n
    sc[0][:6] = [0, 1, pyref.R - 1, (pyref.R - 1) // 2, (pyref.R + 1) // 2, 2]
    S = np.stack([orc.ints_to_mont(col) for col in sc])
    B = zk.Basis(ctx, bases, c)
    got = ctx.msm(B, S)
    assert np.array_equal(got, orc.msm(S, bases))
    B.destroy()


@pytest.mark.parametrize("n,c,n_cols", [(48, 5, 3), (1000, 0, 4), (1000, 7, 20), (8192, 0, 3), (8192, 0, 40), (8192, 13, 5)])
def test_msm_accumulator_form_and_host_normalisation(ctx, monkeypatch, n, c, n_cols):
    """zkfhe_msm_batch_xyzz + zkfhe_g1_xyzz_to_affine (what the prover's commitments go through on one GPU: the MSM stores the
    XYZZ sum, the host normalises a round's points with one inversion) against the oracle MSM, on every tail kernel: k_msm_small
    (K < 64), k_msm_weighted (bucket pipeline, c given), k_msm_table_fold with one partial per visit (<= 16 columns) and with 256;
    a zero column (the identity: ZZ = 0 -> (0, 0)), and the same call through the affine entry point."""
    import zk_fhe_amd as zk
    monkeypatch.setenv("ZKFHE_TABLE_GB", "2")
    rng = np.random.default_rng(7 * n + n_cols)
    bases = _bases(n, seed=n + 3)
    bases[n // 7] = 0
    sc = [[int.from_bytes(rng.bytes(32), "little") % pyref.R for _ in range(n)] for _ in range(3)]
    sc[1] = [int(rng.integers(0, 256)) if i % 3 else (pyref.R - int(rng.integers(1, 1000))) for i in range(n)]
    sc[2] = [0] * n
    S = np.stack([orc.ints_to_mont(sc[j % 3]) for j in range(n_cols)])
    for j in range(3, n_cols):                                       # more columns: the same three kinds, rotated so that they differ
        S[j] = np.roll(S[j], j, axis=0)
    B = zk.Basis(ctx, bases, c)
    aff, raw = ctx.msm_xyzz(B, S)
    want = orc.msm(S, bases)
    assert np.array_equal(aff, want)
    assert np.array_equal(ctx.msm(B, S), want)
    assert not aff[2].any() and not raw[2, 8:].any()                # the zero column: identity, ZZ = ZZZ = 0
    assert raw[0, 8:12].any() and not np.array_equal(raw[0, :8], want[0])   # really the accumulator form (ZZ != 1)
    B.destroy()


@pytest.mark.parametrize("n,n_cols,bits", [(256, 1, 0), (1000, 3, 8), (1000, 3, 9), (8192, 1, 0), (8192, 3, 10), (8192, 4, 11), (8192, 17, 13),
                                           (16384, 2, 8), (65536, 1, 8), (2048, 40, 12), (2048, 300, 10), (2048, 9, 14), (1024, 5, 15)])
def test_msm_table_path(ctx, monkeypatch, n, n_cols, bits):
    """A default basis (window_bits = 0) gets a digit-multiple table and every call against it is a plain sum of table
    points (k_msm_table): forced table widths 8 .. 15 and the budget's own choice, a lone column (butterfly per visit) and
    hundreds (256 partials per visit), non-power-of-two n, an identity base, zero / one / r-1 / (r+-1)/2 scalars, short
    and negative-short columns; called twice (the ticket counters must reset themselves) and against the bucket pipeline
    of the same points (explicit window_bits: no table)."""
    import zk_fhe_amd as zk
    if bits:
        monkeypatch.setenv("ZKFHE_TABLE_BITS", str(bits))
        monkeypatch.setenv("ZKFHE_TABLE_GB", "48")
    else:
        monkeypatch.setenv("ZKFHE_TABLE_GB", "2")
    rng = np.random.default_rng(31 * n + n_cols)
    bases = _bases(n, seed=n + n_cols)
    bases[n // 5] = 0
    sc = [[int.from_bytes(rng.bytes(32), "little") % pyref.R for _ in range(n)] for _ in range(min(n_cols, 4))]
    sc[0][:8] = [0, 1, pyref.R - 1, (pyref.R - 1) // 2, (pyref.R + 1) // 2, 2, 8, pyref.R - 8]
    if n_cols > 1:
        sc[1] = [int(rng.integers(0, 1 << 29)) if i % 3 else (pyref.R - int(rng.integers(1, 1000))) for i in range(n)]
    if n_cols > 2:
        sc[2] = [[0, 1, 536870908][int(rng.integers(0, 3))] for _ in range(n)]
    if n_cols > 3:
        sc[3] = [0] * n
    S = np.stack([orc.ints_to_mont(col) for col in sc])
    if n_cols > 4:   # many columns: the four above, cyclically shifted copies in between
        S = np.stack([np.roll(S[j % 4], j // 4, axis=0) for j in range(n_cols)])
    B = zk.Basis(ctx, bases)
    assert B.has_table
    Bb = zk.Basis(ctx, bases, 11)
    assert not Bb.has_table
    want = ctx.msm(Bb, S)
    assert np.array_equal(want[:4], orc.msm(S[:4], bases))
    for _ in range(2):
        assert np.array_equal(ctx.msm(B, S), want)
    B.destroy()
    Bb.destroy()


@pytest.mark.parametrize("c", [10, 13, 16])
def test_msm_many_columns_all_reduction_paths(ctx, c):
    """More than 16 columns switches the bucket reduction to 8 lanes per marginal sum (64 lanes for the column sums when
    K > 4096) and the task list mixes slice lengths across columns: every column against the oracle."""
    import zk_fhe_amd as zk
    rng = np.random.default_rng(700 + c)
    n, n_cols = 2048, 20
    bases = _bases(n, seed=c)
    S = rand_fr(rng, n_cols * n).reshape(n_cols, n, 4)
    small = [[int(rng.integers(0, 256)) for _ in range(n)] for _ in range(3)]
    for i, col in enumerate(small):
        S[3 * i + 1] = orc.ints_to_mont(col)
    S[7] = orc.ints_to_mont([5] * n)           # one scalar repeated: a few buckets hold every entry (the heavy merge path)
    B = zk.Basis(ctx, bases, c)
    got = ctx.msm(B, S)
    assert np.array_equal(got, orc.msm(S, bases))
    B.destroy()


@pytest.mark.parametrize("c,sort", [(13, "2:6"), (14, "2:7"), (14, "2"), (16, "0"), (16, "2:7"), (16, "1"), (14, "1")])
def test_msm_two_level_sort_geometries(ctx, monkeypatch, c, sort):
    """Bases with K >= 2048 buckets sort their entries in two levels (k_msm_chist / k_msm_cscatter / k_msm_fine): 8 fine bits
    of the bucket id in the entry word unless the index needs the room -- fewer forced here on n = 4096 (ZKFHE_SORT=2:<bits>), and the one-pass sort (ZKFHE_SORT=1) on the
    same inputs; full-width, short, negative-short, constant (one bucket holds a whole window) and zero columns, 24 columns."""
    import zk_fhe_amd as zk
    monkeypatch.setenv("ZKFHE_SORT", sort)
    rng = np.random.default_rng(900 + c)
    n, n_cols = 4096, 24
    bases = _bases(n, seed=40 + c)
    bases[7] = 0
    S = rand_fr(rng, n_cols * n).reshape(n_cols, n, 4)
    S[1] = orc.ints_to_mont([int(rng.integers(0, 256)) if i % 3 else (pyref.R - int(rng.integers(1, 1000))) for i in range(n)])
    S[2] = orc.ints_to_mont([[0, 1, 536870908][int(rng.integers(0, 3))] for _ in range(n)])
    S[3] = orc.ints_to_mont([0] * n)
    S[4] = orc.ints_to_mont([1] * n)
    S[5] = orc.ints_to_mont([pyref.R - 1] * n)
    S[6] = orc.ints_to_mont([(1 << 253) + 12345] * n)
    B = zk.Basis(ctx, bases, c)
    assert not B.has_table
    got = ctx.msm(B, S)
    assert np.array_equal(got, orc.msm(S, bases))
    assert np.array_equal(ctx.msm(B, S[:3]), got[:3])
    B.destroy()


def test_msm_two_level_sort_odd_length(ctx):
    """n = 1001 points, 15-bit windows: 17 017 entries per column -- not a multiple of four, and k_msm_fine reads the staged
    entries in 16-byte loads (the staging stride is rounded up)."""
    import zk_fhe_amd as zk
    rng = np.random.default_rng(77)
    n = 1001
    bases = _bases(n, seed=5)
    S = rand_fr(rng, 5 * n).reshape(5, n, 4)
    S[1] = orc.ints_to_mont([int(rng.integers(0, 256)) for _ in range(n)])
    S[4] = orc.ints_to_mont([3] * n)
    B = zk.Basis(ctx, bases, 15)
    assert not B.has_table
    assert np.array_equal(ctx.msm(B, S), orc.msm(S, bases))
    B.destroy()


def test_msm_k13_batch_linearity(ctx):
    """BASELINE size: 64 columns x 8192 with the witness scalar mix; oracle on a sample + additivity."""
    import zk_fhe_amd as zk
    rng = np.random.default_rng(99)
    n, n_cols = 8192, 64
    bases = _bases(n, seed=13)
    S = rand_fr(rng, n_cols * n).reshape(n_cols, n, 4)
    S[2] = orc.fe_binop("add", S[0], S[1])
    B = zk.Basis(ctx, bases)
    got = ctx.msm(B, S)
    assert np.array_equal(ctx.g1_add(got[0:1], got[1:2])[0], got[2])
    assert np.array_equal(got[5:7], orc.msm(S[5:7], bases))
    B.destroy()


def test_witness_kernels(ctx):
    rng = np.random.default_rng(21)
    Q = 536870909
    n = 1024
    a = rng.integers(0, Q, size=n, dtype=np.uint64)
    b = np.array([[0, 1, Q - 1][int(x)] for x in rng.integers(0, 3, size=n)], dtype=np.uint64)
    prod = ctx.witness_poly_mul_u64(a, b)
    want = [0] * (2 * n - 1)
    ai, bi = [int(x) for x in a], [int(x) for x in b]
    for i in range(n):
        if bi[i]:
            for j in range(n):
                want[i + j] += ai[j] * bi[i]
    assert orc.mont_to_ints(prod) == want
    d, r = ctx.witness_div_mod(prod, Q)
    assert orc.mont_to_ints(d) == [w // Q for w in want]
    assert orc.mont_to_ints(r) == [w % Q for w in want]
    big = orc.ints_to_mont([(1 << 128) - 1, 0, Q, Q - 1, (1 << 127) + 12345])
    d, r = ctx.witness_div_mod(big, (1 << 60) - 93)
    q60 = (1 << 60) - 93
    assert orc.mont_to_ints(d) == [v // q60 for v in [(1 << 128) - 1, 0, Q, Q - 1, (1 << 127) + 12345]]
    assert orc.mont_to_ints(r) == [v % q60 for v in [(1 << 128) - 1, 0, Q, Q - 1, (1 << 127) + 12345]]


def test_msm_point_range_sharding(ctx):
    """an MSM split by point range into per-"rank" partial MSMs, partials added -> the full MSM.
    (One GPU here: the ranks are emulated sequentially; the byte all-gather itself is covered by the gloo test.)"""
    import zk_fhe_amd as zk
    import zk_fhe_amd.batch as B
    rng = np.random.default_rng(77)
    n, n_cols, world = 4096, 3, 4
    bases = _bases(n, seed=5)
    S = rand_fr(rng, n_cols * n).reshape(n_cols, n, 4)
    want = orc.msm(S, bases)
    parts = []
    for r in range(world):
        lo, hi = B.point_range(n, r, world)
        b = zk.Basis(ctx, bases[lo:hi])
        parts.append(ctx.msm(b, np.ascontiguousarray(S[:, lo:hi])))
        b.destroy()
    acc = parts[0]
    for p in parts[1:]:
        acc = ctx.g1_add(acc, p)
    assert np.array_equal(acc, want)
    # world = 1 path of the class
    full = zk.Basis(ctx, bases)
    assert np.array_equal(B.ShardedMsm(ctx, full, 0, 1).msm(S), want)
    full.destroy()


@pytest.mark.parametrize("k", [13, 16])
def test_lookup_permute_matches_halo2_semantics(ctx, k):
    """SURVEY.md 8a row P4 in isolation: k_lookup_permute (k = 16: the row-sliced k_lookup_count / k_lookup_fill of long columns)
    against the oracle's restatement of halo2's
    permute_expression_pair (oracle/halo2_ref.py permute_lookup) -- sorted inputs, table aligned at every first occurrence,
    leftover table values in ascending order -- on uniform, constant, two-valued and saturated columns; an input above 255
    raises the flag."""
    import ctypes
    from oracle import halo2_ref as H
    rng = np.random.default_rng(4)
    n, u = 1 << k, (1 << k) - 107
    cfg = H.Config(k, 1, 1, 1, 1, 109)
    assert cfg.u == u
    table = list(range(256)) + [0] * (n - 256)
    cols = [[int(v) for v in rng.integers(0, 256, u)],
            [7] * u,
            [int(v) for v in rng.choice([0, 255], u)],
            [i % 256 for i in range(u)],
            [0] * (u - 1) + [255]]
    S = np.stack([orc.ints_to_mont(c + [123456789] * (n - u)) for c in cols])   # rows >= u hold junk: they must be ignored
    d = ctx.to_device(S)
    a, s = ctx.alloc(len(cols) * n * 32), ctx.alloc(len(cols) * n * 32)
    vp = ctypes.c_void_p
    ctx.lib.zkfhe_lookup_permute.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint32, vp, vp, ctypes.POINTER(ctypes.c_int)]
    flag = ctypes.c_int(-1)
    ctx._check(ctx.lib.zkfhe_lookup_permute(ctx.h, d.at(0), len(cols), n, u, a.at(0), s.at(0), ctypes.byref(flag)))
    assert flag.value == 0
    ga, gs = a.download(shape=(len(cols), n, 4)), s.download(shape=(len(cols), n, 4))
    for i, c in enumerate(cols):
        wa, ws = H.permute_lookup(cfg, c, table)
        assert orc.mont_to_ints(ga[i][:u]) == wa and orc.mont_to_ints(gs[i][:u]) == ws, "column %d" % i
    bad = list(cols[0])
    bad[100] = 256
    d2 = ctx.to_device(orc.ints_to_mont(bad + [0] * (n - u))[None])
    ctx._check(ctx.lib.zkfhe_lookup_permute(ctx.h, d2.at(0), 1, n, u, a.at(0), s.at(0), ctypes.byref(flag)))
    assert flag.value == 1
    for b in (d, d2, a, s):
        b.free()


def test_msm_sparse_terms(ctx):
    """zkfhe_msm_sparse (one wave per output over the digit-multiple table) against the oracle MSM of the same cells as full
    columns: random rows, zero / one / r - 1 scalars, several cells per slot, an empty slot."""
    import ctypes
    import zk_fhe_amd as zk
    rng = np.random.default_rng(77)
    n, n_slots = 4096, 5
    bases = _bases(n, seed=3)
    B = zk.Basis(ctx, bases)
    cells = [(int(rng.integers(0, n)), int(rng.integers(0, 4)), int.from_bytes(rng.bytes(32), "little") % pyref.R) for _ in range(14)]
    cells += [(7, 0, 0), (8, 1, 1), (9, 2, pyref.R - 1), (7, 3, 5)]      # slot 4 stays empty
    cols = [[0] * n for _ in range(n_slots)]
    for row, slot, v in cells:
        cols[slot][row] = (cols[slot][row] + v) % pyref.R
    want = orc.msm(np.stack([orc.ints_to_mont(c) for c in cols]), bases)

    class Term(ctypes.Structure):
        _fields_ = [("scalar", ctypes.c_uint64 * 4), ("row", ctypes.c_uint32), ("slot", ctypes.c_uint32)]
    arr = (Term * len(cells))()
    mont = orc.ints_to_mont([v for _, _, v in cells])
    for i, (row, slot, _) in enumerate(cells):
        for j in range(4):
            arr[i].scalar[j] = int(mont[i][j])
        arr[i].row, arr[i].slot = row, slot
    raw = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
    d = ctx.to_device(raw)
    out = ctx.alloc(n_slots * 64)
    vp = ctypes.c_void_p
    ctx.lib.zkfhe_msm_sparse.argtypes = [vp, vp, vp, ctypes.c_size_t, ctypes.c_size_t, vp]
    assert ctx.lib.zkfhe_basis_has_multiples(B.h) == 1
    ctx._check(ctx.lib.zkfhe_msm_sparse(ctx.h, B.h, d.at(0), len(cells), n_slots, out.at(0)))
    assert np.array_equal(out.download(shape=(n_slots, 8)), want)
    d.free(), out.free()
    B.destroy()


def test_2_13_tile_against_the_oracle(ctx):
    """The 2^13 tile (k_ntt13: every column transform of a k = 13 proof, the tile of the longer rows): forward, inverse (with its
    n^-1), out of place, 37 columns (ragged groups of eight), in place through scratch, the coset extension with 2 and 1 extension
    bits and the extension + inverse round trip, all against the C oracle.  (Round 5 ran this for two kernels in interpreters of
    their own; the quarter-column kernel that lost the comparison is tools/exp/patches/ntt13_quarter.patch, test included.)"""
    rng = np.random.default_rng(13)
    n, n_cols = 8192, 37
    a = orc.ints_to_mont([int.from_bytes(rng.bytes(32), "little") % pyref.R for _ in range(3 * n)]).reshape(3, n, 4)
    a = np.concatenate([a] * 13)[:n_cols].copy()
    a[5] = np.roll(a[5], 7, axis=0)
    for inv in (False, True):
        src, dst = ctx.to_device(a), ctx.alloc(n_cols * n * 32)
        ctx.ntt_to_dev(src, dst, n_cols, 13, inverse=inv)
        got = dst.download(shape=(n_cols, n, 4))
        src.free(), dst.free()
        want = orc.ntt(a[:6], 13, inv)
        assert np.array_equal(got[:6], want), "ntt inverse=%s" % inv
        assert np.array_equal(got[6:9], want[:3]) and np.array_equal(got[36], want[36 % 3])
        assert np.array_equal(ctx.ntt(a[:2], 13, inv), want[:2])          # in place (through scratch)
    g = orc.ints_to_mont([pyref.FR_GEN])[0]
    for lef in (2, 1):
        E = 1 << lef
        got = ctx.coset_ntt(a[:2], 13, lef, g)
        for c in range(2):
            nat = orc.coset_ntt(a[c], 13 + lef, g)
            assert np.array_equal(got[c], nat.reshape(n, E, 4).transpose(1, 0, 2).reshape(n * E, 4)), "coset lef=%d" % lef
        assert np.array_equal(ctx.coset_ntt(got, 13, lef, g, inverse=True)[:, :n], a[:2])
