#!/usr/bin/env python3
"""Generates tests/golden/bn254_vectors.json from exact big-integer arithmetic (oracle/pyref.py).

Run from the repo root:  python tests/golden/gen_vectors.py
The vectors pin the C oracle (oracle/oracle.c); the C oracle then checks the HIP kernels at sizes
Python cannot reach.  Values are canonical integers as hex strings (no Montgomery form in the file).
"""
import json
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import pyref as P  # noqa: E402


def h(x):
    return hex(x)


def main():
    rng = random.Random(20240613)
    out = {"constants": {
        "fr_modulus": h(P.R), "fq_modulus": h(P.Q),
        "fr_root_of_unity": h(P.FR_ROOT_OF_UNITY), "fr_delta": h(P.FR_DELTA), "fr_s": P.FR_S,
        "fr_R": h(P.MONT % P.R), "fq_R": h(P.MONT % P.Q),
    }}
    for name, p in (("fr", P.R), ("fq", P.Q)):
        edge = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, (1 << 253) % p, (1 << 64) - 1, 1 << 128]
        a = edge + [rng.randrange(p) for _ in range(54)]
        b = list(reversed(edge)) + [rng.randrange(p) for _ in range(54)]
        out[name] = {
            "a": [h(x) for x in a], "b": [h(x) for x in b],
            "add": [h((x + y) % p) for x, y in zip(a, b)],
            "sub": [h((x - y) % p) for x, y in zip(a, b)],
            "mul": [h((x * y) % p) for x, y in zip(a, b)],
            "inv": [h(pow(x, -1, p) if x else 0) for x in a],
        }
    # G1
    g = P.G1_GEN
    ks = [1, 2, 3, P.R - 1, P.R - 2, (1 << 128) + 7] + [rng.randrange(P.R) for _ in range(10)]
    pts = [P.g1_mul(g, k) for k in ks]
    assert all(P.g1_is_on_curve(x) for x in pts)
    adds = []
    cases = [(0, 1), (1, 1), (0, 3), (3, 0), (5, 6), (7, 7), (8, 9)]  # incl. doubling and P + (-P)
    for i, j in cases:
        s = P.g1_add(pts[i], pts[j])
        adds.append({"i": i, "j": j, "sum": [h(v) for v in P.g1_affine_to_xy(s)]})
    ident = P.g1_add(pts[0], None)
    adds.append({"i": 0, "j": -1, "sum": [h(v) for v in P.g1_affine_to_xy(ident)]})
    out["g1"] = {"k": [h(k) for k in ks], "kG": [[h(v) for v in P.g1_affine_to_xy(x)] for x in pts], "add": adds}
    # small MSMs (naive double-and-add)
    msms = []
    for n in (1, 2, 5, 16, 33):
        bases = [P.g1_mul(g, rng.randrange(1, P.R)) for _ in range(n)]
        sc = [rng.randrange(P.R) for _ in range(n)]
        if n >= 5:
            sc[0] = 0
            sc[1] = 1
            sc[2] = P.R - 1
            sc[3] = 255
            bases[4] = None  # identity base
        res = P.g1_msm(sc, bases)
        msms.append({"scalars": [h(x) for x in sc], "bases": [[h(v) for v in P.g1_affine_to_xy(b)] for b in bases],
                     "result": [h(v) for v in P.g1_affine_to_xy(res)]})
    out["msm"] = msms
    # NTT: naive O(n^2) DFT vs halo2 omega; and round trip
    ntts = []
    for log_n in (1, 2, 3, 4, 6):
        n = 1 << log_n
        a = [rng.randrange(P.R) for _ in range(n)]
        w = P.root_of_unity(log_n)
        fwd = P.ntt_naive(a, w)
        assert P.best_fft(list(a), w, log_n) == fwd
        assert P.ifft(list(fwd), log_n) == a
        ntts.append({"log_n": log_n, "omega": h(w), "in": [h(x) for x in a], "out": [h(x) for x in fwd]})
    out["ntt"] = ntts
    path = os.path.join(os.path.dirname(__file__), "bn254_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
