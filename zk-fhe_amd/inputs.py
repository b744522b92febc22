"""BFV key generation + encryption for arbitrary (N, Q, T, B): the input files of the circuit.

The reference's README (README.md:25) points at an external Python generator for `data/bfv/*.in`; this module is the
in-tree replacement, so configurations 3-5 of BASELINE.json (batches of k = 13 inputs, N = 4096 with a 60-bit Q,
N = 16384) can be produced without it.  Output = the `CircuitInput` JSON of examples/bfv.rs:50-61: nine lists of decimal
strings, coefficients in big-endian order (highest degree first), every value reduced to [0, Q).

    sk  s  <- ternary {-1, 0, 1}^N            pk = (pk0, pk1) = (-(a s + e) mod Q, a),  a uniform, e ~ chi_error
    enc(m): u <- ternary, e0, e1 ~ chi_error   c0 = pk0 u + floor(Q/T) m + e0,  c1 = pk1 u + e1     in Z_Q[x]/(x^N + 1)

chi_error is a discrete Gaussian (sigma 3.2) clipped to [-B, B] (the circuit range-checks e0, e1 against B,
examples/bfv.rs:189-200).  All products have a ternary factor, so they are sums / differences of negacyclic rotations:
exact in int64 for any Q < 2^62.

    python -m zk_fhe_amd.inputs --n 1024 --q 536870909 --t 7 --b 19 --seed 1 > data/bfv/my.in
"""
import argparse
import json
import sys

import numpy as np


def _negacyclic_ternary(a, s, q):
    """a * s in Z_q[x]/(x^N + 1) for s in {-1, 0, 1}^N; little-endian coefficient arrays (index = degree)."""
    n = len(a)
    out = np.zeros(n, dtype=np.int64)
    for i in np.nonzero(s)[0]:
        sh = np.empty(n, dtype=np.int64)
        sh[i:] = a[: n - i]
        sh[:i] = (q - a[n - i:]) % q
        out = (out + (sh if s[i] == 1 else (q - sh) % q)) % q
    return out


def _chi_error(rng, n, b, sigma=3.2):
    return np.clip(np.rint(rng.normal(0.0, sigma, n)), -b, b).astype(np.int64)


def keygen(n, q, b, rng):
    """-> (sk, (pk0, pk1)): little-endian int64 arrays; sk in {-1,0,1}, pk in [0, q)."""
    s = rng.integers(-1, 2, n, dtype=np.int64)
    a = rng.integers(0, q, n, dtype=np.int64)
    e = _chi_error(rng, n, b)
    pk0 = (q - (_negacyclic_ternary(a, s, q) + e) % q) % q
    return s, (pk0, a)


def encrypt(pk, m, q, t, b, rng):
    """m: plaintext coefficients centred in (-t/2, t/2].  -> dict with u, e0, e1, c0, c1 (little-endian int64)."""
    pk0, pk1 = pk
    n = len(pk0)
    u = rng.integers(-1, 2, n, dtype=np.int64)
    e0, e1 = _chi_error(rng, n, b), _chi_error(rng, n, b)
    delta = q // t
    md = np.array([(int(x) % q) * delta % q for x in m], dtype=np.int64)
    c0 = ((_negacyclic_ternary(pk0, u, q) + md) % q + e0) % q
    c1 = (_negacyclic_ternary(pk1, u, q) + e1) % q
    return dict(u=u, e0=e0, e1=e1, c0=c0, c1=c1)


def decrypt(sk, c0, c1, q, t):
    """round(t/q * (c0 + c1 s)) mod t, centred.  Used by the self-test of the generator."""
    n = len(c0)
    v = (c0 + _negacyclic_ternary(c1, sk, q)) % q
    out = np.empty(n, dtype=np.int64)
    for i in range(n):
        x = int(v[i])
        if x > q // 2:
            x -= q
        r = (2 * t * x + q) // (2 * q)  # round(t x / q)
        r %= t
        out[i] = r - t if r > t // 2 else r
    return out


def generate(n=1024, q=536870909, t=7, b=19, seed=0, with_secret=False, key_seed=None):
    """One circuit input as a dict of lists of decimal strings (big-endian), i.e. what json.dumps turns into a *.in file.
    key_seed: draw the key pair from its own stream, so that different `seed`s give different encryptions under ONE public key
    (the usual shape of a batch: one recipient, many messages); None = a fresh key per input from the same stream."""
    rng = np.random.default_rng(seed)
    sk, pk = keygen(n, q, b, rng if key_seed is None else np.random.default_rng(key_seed))
    m = rng.integers(-(t // 2), t // 2 + 1, n, dtype=np.int64)
    ct = encrypt(pk, m, q, t, b, rng)
    cyclo = np.zeros(n + 1, dtype=np.int64)
    cyclo[0] = cyclo[n] = 1

    def fmt(v):
        return [str(int(x) % q) for x in v[::-1]]
    out = dict(pk0=fmt(pk[0]), pk1=fmt(pk[1]), m=fmt(m), u=fmt(ct["u"]), e0=fmt(ct["e0"]), e1=fmt(ct["e1"]),
               c0=fmt(ct["c0"]), c1=fmt(ct["c1"]), cyclo=fmt(cyclo))
    if with_secret:
        return out, dict(sk=sk, m=m, c0=ct["c0"], c1=ct["c1"])
    return out


def config3_vector(seed, n=1024, q=536870909, t=7, b=19):
    """One vector of SURVEY.md section 8(d) config 3 as CircuitInput JSON text: rng = numpy.random.default_rng(seed); pk0, pk1 uniform
    in [0, Q), u uniform in {0, 1, Q-1}, m uniform centred mod T, e0 / e1 a discrete Gaussian (sigma 3.2) clipped to +-B, and
    c0 = pk0 u + floor(Q/T) m + e0, c1 = pk1 u + e1 in Z_Q[x]/(x^N + 1) (formula checked against data/bfv/bfv.in, KAT 1).
    Every vector has its own public key; coefficients big-endian as in the reference's files."""
    rng = np.random.default_rng(seed)
    pk0 = rng.integers(0, q, n, dtype=np.int64)
    pk1 = rng.integers(0, q, n, dtype=np.int64)
    u = rng.choice(np.array([0, 1, q - 1], dtype=np.int64), n)
    m = rng.choice(np.array(list(range(0, t // 2 + 1)) + [q - i for i in range(1, t // 2 + 1)], dtype=np.int64), n)
    e = np.clip(np.rint(rng.normal(0, 3.2, (2, n))), -b, b).astype(np.int64) % q

    def negacyclic(a, s):  # big-endian coefficient order in, out; s taken as residues in [0, q)
        a, s = a[::-1], s[::-1]
        out = np.zeros(n, dtype=np.int64)
        for i in np.nonzero(s)[0]:
            sh = np.empty(n, dtype=np.int64)
            sh[i:] = a[: n - i]
            sh[:i] = (q - a[n - i:]) % q
            out = (out + (sh * int(s[i])) % q) % q
        return out[::-1]
    c0 = (negacyclic(pk0, u) + (q // t) * m % q + e[0]) % q
    c1 = (negacyclic(pk1, u) + e[1]) % q
    cyclo = np.zeros(n + 1, dtype=np.int64)
    cyclo[0] = cyclo[n] = 1
    s = lambda v: [str(int(x)) for x in v]  # noqa: E731
    return json.dumps(dict(pk0=s(pk0), pk1=s(pk1), m=s(m), u=s(u), e0=s(e[0]), e1=s(e[1]), c0=s(c0), c1=s(c1), cyclo=s(cyclo)))


def config3_batch(bfv_in_text, count=64, base_seed=20240613):
    """BASELINE configs[2] / SURVEY.md section 8(d) config 3: the reference's data/bfv/bfv.in followed by count - 1 seeded vectors
    (seeds base_seed + 1 ... base_seed + count - 1), as a list of JSON texts (bytes)."""
    first = bfv_in_text if isinstance(bfv_in_text, bytes) else bfv_in_text.encode()
    return [first] + [config3_vector(base_seed + i).encode() for i in range(1, count)]


def empty(n=1024):
    """The all-zero input the reference uses for keygen (data/bfv/bfv_empty.in)."""
    return {k: ["0"] * (n + 1 if k == "cyclo" else n) for k in ("pk0", "pk1", "m", "u", "e0", "e1", "c0", "c1", "cyclo")}


def main(argv=None):
    ap = argparse.ArgumentParser(description="BFV circuit input generator (CircuitInput JSON on stdout)")
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--q", type=int, default=536870909)
    ap.add_argument("--t", type=int, default=7)
    ap.add_argument("--b", type=int, default=19)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--key-seed", type=int, default=None, help="seed of the key pair alone: different --seed values then encrypt under one public key")
    ap.add_argument("--empty", action="store_true", help="all-zero input (keygen)")
    a = ap.parse_args(argv)
    if a.q >= 1 << 62:
        ap.error("q must be below 2^62")
    json.dump(empty(a.n) if a.empty else generate(a.n, a.q, a.t, a.b, a.seed, key_seed=a.key_seed), sys.stdout)
    sys.stdout.write("\n")


if __name__ == "__main__":
    main()
