"""TEST INFRASTRUCTURE (oracle).  The 32-byte compressed form of a BN254 G1 point in the proof byte stream: the one
definition both oracle transcripts (halo2_ref.py, poseidon_ref.py) use.  The product's twin is
zk-fhe_amd/host/point_encoding.hpp; tests/test_point_encoding.py pins both to tests/golden/point_encoding.json.

Layout: halo2curves `new_curve_impl!` GroupEncoding for bn256 G1Affine (0.3.2 .. 0.5 line / halo2curves-axiom, the
crate behind the `Snark.proof` bytes of reference examples/bfv.rs:311):
    to_bytes:   identity -> 31 zero bytes, then 0b1000_0000;  else x little-endian, byte 31 |= (y & 1) << 6
    from_bytes: is_inf = byte31 >> 7, ysign = (byte31 >> 6) & 1, byte31 &= 0b0011_1111
Parity unpinned: no reference-made proof exists in /root/reference and the crate is not on this machine."""

import os

Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583

# ONE switch shared with the product (zk-fhe_amd/host/point_encoding.hpp): ZKFHE_POINT_ENCODING = "halo2curves-0.3.2" (default)
# or "halo2curves-0.3.1" (sign in bit 7, identity = 32 zero bytes)
if os.environ.get("ZKFHE_POINT_ENCODING") == "halo2curves-0.3.1":
    SIGN_BIT, IDENTITY_BIT, X_MASK = 0x80, 0x00, 0x7F
else:
    SIGN_BIT = 0x40       # bit 6 of byte 31: y is odd
    IDENTITY_BIT = 0x80   # bit 7 of byte 31: the point at infinity (every other bit zero)
    X_MASK = 0x3F


def point_compress(P):
    if P is None:
        b = bytearray(32)
        b[31] |= IDENTITY_BIT
        return bytes(b)
    x, y = P
    b = bytearray(x.to_bytes(32, "little"))
    if y & 1:
        b[31] |= SIGN_BIT
    return bytes(b)


def point_decompress(b):
    b = bytearray(b)
    if (b[31] & IDENTITY_BIT) if IDENTITY_BIT else not any(b):
        assert not any(b[:31]) and b[31] == IDENTITY_BIT, "non-canonical identity encoding"
        return None
    sign = 1 if b[31] & SIGN_BIT else 0
    b[31] &= X_MASK
    x = int.from_bytes(b, "little")
    assert x < Q, "point x not reduced"
    y2 = (x * x * x + 3) % Q
    y = pow(y2, (Q + 1) // 4, Q)
    assert y * y % Q == y2, "not on curve"
    if (y & 1) != sign:
        y = Q - y
    return (x, y)
