"""Pins the C oracle (oracle/oracle.c) against exact big-integer golden vectors
(tests/golden/bn254_vectors.json, made by tests/golden/gen_vectors.py) and against the
reference-pinned constants (SURVEY.md section 4 KAT 4: F::MODULUS at src/poly_chip.rs:90)."""
import json
import os

import numpy as np
import pytest

from oracle import binding as orc
from oracle import pyref

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def vec():
    with open(os.path.join(HERE, "golden", "bn254_vectors.json")) as f:
        return json.load(f)


def hx(lst):
    return [int(x, 16) for x in lst]


def test_constants(vec):
    c = vec["constants"]
    assert int(c["fr_modulus"], 16) == 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
    assert int(c["fr_modulus"], 16).bit_length() == 254
    assert pow(pyref.FR_ROOT_OF_UNITY, 1 << 28, pyref.R) == 1
    assert pow(pyref.FR_ROOT_OF_UNITY, 1 << 27, pyref.R) != 1
    assert hex(pyref.FR_ROOT_OF_UNITY).startswith("0x3ddb9f5") and hex(pyref.FR_ROOT_OF_UNITY).endswith("c37c9c")
    assert hex(pyref.FR_DELTA).startswith("0x9226b6e") and hex(pyref.FR_DELTA).endswith("33e9a2")
    # Montgomery one of the oracle == R mod p
    one = orc.ints_to_mont([1], 0)
    assert orc.arr_to_ints(one)[0] == int(c["fr_R"], 16)
    one = orc.ints_to_mont([1], 1)
    assert orc.arr_to_ints(one)[0] == int(c["fq_R"], 16)
    assert orc.mont_to_ints(orc.root_of_unity(28))[0] == pyref.FR_ROOT_OF_UNITY
    assert orc.mont_to_ints(orc.root_of_unity(13))[0] == pyref.root_of_unity(13)


@pytest.mark.parametrize("name,which", [("fr", 0), ("fq", 1)])
def test_field_ops(vec, name, which):
    v = vec[name]
    a = orc.ints_to_mont(hx(v["a"]), which)
    b = orc.ints_to_mont(hx(v["b"]), which)
    assert orc.mont_to_ints(a, which) == hx(v["a"])
    for op in ("add", "sub", "mul"):
        assert orc.mont_to_ints(orc.fe_binop(op, a, b, which), which) == hx(v[op]), op
    assert orc.mont_to_ints(orc.fe_inv(a, which), which) == hx(v["inv"])
    if which == 0:
        assert orc.mont_to_ints(orc.fr_batch_inv(a), 0) == hx(v["inv"])


def test_g1(vec):
    g = vec["g1"]
    G = orc.points_to_arr([pyref.G1_GEN] * len(g["k"]))
    k = orc.ints_to_mont(hx(g["k"]), 0)
    got = orc.arr_to_points(orc.g1_mul(G, k))
    want = [None if (int(x, 16) == 0 and int(y, 16) == 0) else (int(x, 16), int(y, 16)) for x, y in g["kG"]]
    assert got == want
    pts = orc.points_to_arr(want)
    for p in pts:
        assert orc.g1_on_curve(p)
    for case in g["add"]:
        a = pts[case["i"]]
        b = pts[case["j"]] if case["j"] >= 0 else np.zeros(8, dtype=np.uint64)
        s = orc.arr_to_points(orc.g1_add(a, b))[0]
        w = (int(case["sum"][0], 16), int(case["sum"][1], 16))
        assert (s or (0, 0)) == w


def test_msm(vec):
    for m in vec["msm"]:
        bases = [None if (int(x, 16) == 0 and int(y, 16) == 0) else (int(x, 16), int(y, 16)) for x, y in m["bases"]]
        B = orc.points_to_arr(bases)
        S = orc.ints_to_mont(hx(m["scalars"]), 0)
        want = (int(m["result"][0], 16), int(m["result"][1], 16))
        assert (orc.arr_to_points(orc.msm_naive(S, B))[0] or (0, 0)) == want
        assert (orc.arr_to_points(orc.msm(S[None], B))[0] or (0, 0)) == want


def test_ntt(vec):
    for t in vec["ntt"]:
        log_n = t["log_n"]
        a = orc.ints_to_mont(hx(t["in"]), 0)
        got = orc.mont_to_ints(orc.ntt(a[None], log_n, False)[0], 0)
        assert got == hx(t["out"])
        back = orc.mont_to_ints(orc.ntt(orc.ints_to_mont(hx(t["out"]))[None], log_n, True)[0], 0)
        assert back == hx(t["in"])
        w = orc.ints_to_mont([int(t["omega"], 16)])
        assert orc.mont_to_ints(orc.fft(a, log_n, w)) == hx(t["out"])


def test_msm_pippenger_vs_naive_medium():
    rng = np.random.default_rng(7)
    n = 300
    start = orc.ints_to_mont([12345])
    step = orc.ints_to_mont([987654321])
    bases = orc.g1_powers(start, step, n)
    sc = [int.from_bytes(rng.bytes(32), "little") % pyref.R for _ in range(n)]
    sc[:4] = [0, 1, pyref.R - 1, 255]
    S = orc.ints_to_mont(sc)
    assert np.array_equal(orc.msm(S[None], bases)[0], orc.msm_naive(S, bases))


def test_coset_roundtrip():
    rng = np.random.default_rng(3)
    n, log_ext = 16, 6
    a = orc.ints_to_mont([int.from_bytes(rng.bytes(32), "little") % pyref.R for _ in range(n)])
    g = orc.ints_to_mont([pyref.FR_GEN])
    ext = orc.coset_ntt(a, log_ext, g)
    # direct evaluation at g*w^i
    w = pyref.root_of_unity(log_ext)
    ai = orc.mont_to_ints(a)
    for i in (0, 1, 5, 63):
        x = pyref.FR_GEN * pow(w, i, pyref.R) % pyref.R
        want = sum(c * pow(x, j, pyref.R) for j, c in enumerate(ai)) % pyref.R
        assert orc.mont_to_ints(ext[i])[0] == want
    back = orc.coset_ntt(ext, log_ext, g, inverse=True)
    assert orc.mont_to_ints(back[:n]) == ai
    assert all(v == 0 for v in orc.mont_to_ints(back[n:]))
