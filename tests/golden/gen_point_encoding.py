"""Writes tests/golden/point_encoding.json: BN254 G1 points and their 32-byte compressed form, spelled out here bit by bit
(NOT through oracle/point_encoding.py, which the vectors are there to pin) following halo2curves' bn256 `to_bytes`
(0.3.2 .. 0.5 / halo2curves-axiom): x little-endian, (y & 1) << 6 into byte 31, identity = 0x80 in byte 31 and zeros.

    python tests/golden/gen_point_encoding.py
"""
import json
import os

Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583


def add(P, R):
    if P is None:
        return R
    if R is None:
        return P
    (x1, y1), (x2, y2) = P, R
    if x1 == x2:
        if (y1 + y2) % Q == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, Q) % Q
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, Q) % Q
    x3 = (lam * lam - x1 - x2) % Q
    return (x3, (lam * (x1 - x3) - y1) % Q)


def mul(k, P):
    acc = None
    while k:
        if k & 1:
            acc = add(acc, P)
        P = add(P, P)
        k >>= 1
    return acc


def enc(P):
    out = [0] * 32
    if P is None:
        out[31] = 0b1000_0000
        return bytes(out)
    x, y = P
    for i in range(32):
        out[i] = (x >> (8 * i)) & 0xFF
    assert out[31] < 0b0100_0000          # q < 2^254: the two top bits are free
    out[31] |= (y & 1) << 6
    return bytes(out)


G = (1, 2)
cases = []
for k in [1, 2, 3, 5, 7, 0xDEADBEEF, 2**200 + 12345, Q - 5]:
    P = mul(k, G)
    for pt in (P, (P[0], Q - P[1])):
        cases.append({"x": hex(pt[0]), "y": hex(pt[1]), "bytes": enc(pt).hex()})
cases.append({"x": None, "y": None, "bytes": enc(None).hex()})
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "point_encoding.json"), "w") as f:
    json.dump({"layout": {"sign_bit": 6, "identity_bit": 7, "x_mask_byte31": 0x3F}, "cases": cases}, f, indent=1)
print(len(cases), "cases")
