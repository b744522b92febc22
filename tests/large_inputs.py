"""The inputs of the large configurations (BASELINE configs[3] / [4]: N = 4096 / 16384, Q = 2^60 - 93), shared by the GPU tests
(tests/test_gpu_prover.py) and by the generator of the oracle's golden digests (tests/golden/gen_large_proof_digests.py)."""
import numpy as np

Q60, T, B = (1 << 60) - 93, 7, 19


def large_input(N, seed=4):
    """A valid BFV encryption at ring degree N under the 60-bit modulus, as the CircuitInput dict (big-endian decimal strings)."""
    Q = Q60
    rng = np.random.default_rng(seed)
    pk0 = rng.integers(0, Q, N, dtype=np.int64)
    pk1 = rng.integers(0, Q, N, dtype=np.int64)
    u = rng.choice(np.array([0, 1, -1], dtype=np.int64), N)
    m = rng.choice(np.array([0, 1, 2, 3, -1, -2, -3], dtype=np.int64), N)
    e = np.clip(np.rint(rng.normal(0, 3.2, (2, N))), -B, B).astype(np.int64)

    def negacyclic_pm1(a, s):  # a * s in Z_Q[x]/(x^N+1), s in {0,+1,-1}; big-endian in and out
        a, s = a[::-1], s[::-1]
        out = np.zeros(N, dtype=np.int64)
        for i in np.nonzero(s)[0]:
            sh = np.empty(N, dtype=np.int64)
            sh[i:] = a[: N - i]
            sh[:i] = (Q - a[N - i:]) % Q
            out = (out + (sh if s[i] == 1 else (Q - sh) % Q)) % Q
        return out[::-1]
    delta = Q // T
    md = np.array([(int(x) % Q) * delta % Q for x in m], dtype=np.int64)
    c0 = (negacyclic_pm1(pk0, u) + md) % Q
    c0 = (c0 + e[0] % Q) % Q
    c1 = (negacyclic_pm1(pk1, u) + e[1] % Q) % Q
    s = lambda v: [str(int(x) % Q) for x in v]  # noqa: E731
    return dict(pk0=s(pk0), pk1=s(pk1), m=s(m), u=s(u), e0=s(e[0]), e1=s(e[1]), c0=s(c0), c1=s(c1), cyclo=s([1] + [0] * (N - 1) + [1]))
