"""TEST INFRASTRUCTURE (oracle): ChaCha20 block function, restated from RFC 7539 section 2.3, and the secret of the reference's
unsafe test SRS.

The reference's setup is third-party code (halo2-scaffold `gen_srs(k)`, reached from examples/bfv.rs:311 and described by
README.md:34): `ParamsKZG::<Bn256>::setup(k, ChaCha20Rng::from_seed(Default::default()))`, whose `s = Fr::random(rng)` is
halo2curves' `Fr::from_u512([rng.next_u64(); 8])` -- the first 64 keystream bytes of the all-zero key / counter 0 / stream 0,
little-endian, reduced mod r.  Pinned by the published zero-key keystream and by RFC 7539's own block-function vector
(tests/test_srs_file.py)."""
import struct

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def _rotl(v, n):
    return ((v << n) & 0xFFFFFFFF) | (v >> (32 - n))


def block(key, counter_nonce):
    """key: 32 bytes; counter_nonce: the four state words 12..15"""
    s = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(struct.unpack("<8I", bytes(key))) + [int(w) & 0xFFFFFFFF for w in counter_nonce]
    x = list(s)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & 0xFFFFFFFF
        x[d] = _rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & 0xFFFFFFFF
        x[b] = _rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & 0xFFFFFFFF
        x[d] = _rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & 0xFFFFFFFF
        x[b] = _rotl(x[b] ^ x[c], 7)
    for _ in range(10):
        qr(0, 4, 8, 12), qr(1, 5, 9, 13), qr(2, 6, 10, 14), qr(3, 7, 11, 15)
        qr(0, 5, 10, 15), qr(1, 6, 11, 12), qr(2, 7, 8, 13), qr(3, 4, 9, 14)
    return struct.pack("<16I", *[(a + b) & 0xFFFFFFFF for a, b in zip(x, s)])


def reference_srs_secret():
    """s of ParamsKZG::setup(k, ChaCha20Rng::from_seed([0; 32])): the same for every k"""
    return int.from_bytes(block(bytes(32), (0, 0, 0, 0)), "little") % R
