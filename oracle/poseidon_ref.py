"""ORACLE (test infrastructure only): Poseidon over BN254 Fr as snark-verifier's `PoseidonTranscript<NativeLoader>` uses it.

What this restates (third-party, not on disk -- SURVEY.md section 8c / Appendix B "Transcript"):
  * the `poseidon` crate's `Spec::new(R_F, R_P)` (privacy-scaling-explorations/poseidon, the dependency of
    snark-verifier): round constants and the Cauchy MDS matrix from the Grain LFSR of the Poseidon paper
    (eprint 2019/458, appendix F; reference script `generate_parameters_grain.sage`), field tag 1 (prime), S-box tag 0
    (x^alpha, alpha = 5), n = 254 bits, t = 3, R_F = 8, R_P = 57 -- snark-verifier-sdk's T = 3, RATE = 2, R_F = 8,
    R_P = 57 (the reference reaches it through `gen_snark_shplonk`, examples/bfv.rs:311, Cargo.toml:9 "aggregation");
  * snark-verifier `util/hash/poseidon.rs`: sponge state [2^64, 0, 0], `update` buffers, `squeeze` absorbs the
    buffer in chunks of RATE (a chunk shorter than RATE is followed by a 1 in the next state word; an exact multiple
    gets one more permutation with an empty chunk) and returns state[1];
  * snark-verifier `system/halo2/transcript/halo2.rs`: a scalar is absorbed as itself, a G1 point as its two affine
    coordinates mapped Fq -> Fr (integer value mod r), the byte stream carries 32-byte compressed points and 32-byte
    little-endian scalars.
The permutation here is the PLAIN Hades permutation (add round constants, S-box, MDS); the crate's optimised form
(pre-sparse MDS) is an equivalent rewriting and yields the same output.

Pinned by the public known-answer vector of the Poseidon reference implementation for exactly this instance
(`poseidonperm_x5_254_3`: permutation of [0, 1, 2]) and by the first round constant / MDS entry that circomlib's
`poseidon_constants` also carries -- tests/test_poseidon.py, tests/golden/poseidon_bn254_t3.json.
"""
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
T, RATE, R_F, R_P, ALPHA = 3, 2, 8, 57, 5


class Grain:
    """80-bit LFSR of the Poseidon parameter generator: x^80 + x^62 + x^51 + x^38 + x^23 + x^13 + 1, self-shrinking output."""

    def __init__(self, field_bits=254, t=T, r_f=R_F, r_p=R_P, field_tag=1, sbox_tag=0):
        bits = []

        def put(value, width):
            bits.extend((value >> (width - 1 - i)) & 1 for i in range(width))  # most significant bit first
        put(field_tag, 2)
        put(sbox_tag, 4)
        put(field_bits, 12)
        put(t, 12)
        put(r_f, 10)
        put(r_p, 10)
        put((1 << 30) - 1, 30)
        assert len(bits) == 80
        self.s = bits
        self.field_bits = field_bits
        for _ in range(160):
            self._step()

    def _step(self):
        s = self.s
        b = s[62] ^ s[51] ^ s[38] ^ s[23] ^ s[13] ^ s[0]
        s.pop(0)
        s.append(b)
        return b

    def next_bit(self):
        # evaluate bits in pairs: a leading 1 outputs the second bit, a leading 0 discards it
        while True:
            a = self._step()
            b = self._step()
            if a:
                return b

    def next_int(self):
        v = 0
        for _ in range(self.field_bits):
            v = (v << 1) | self.next_bit()   # first bit is the most significant
        return v

    def next_field_element(self):
        while True:
            v = self.next_int()
            if v < R:
                return v

    def next_field_element_without_rejection(self):
        return self.next_int() % R


def generate_constants():
    """(round_constants [R_F + R_P][T], mds [T][T]) -- python ints"""
    g = Grain()
    rc = [[g.next_field_element() for _ in range(T)] for _ in range(R_F + R_P)]
    xs = [g.next_field_element_without_rejection() for _ in range(T)]
    ys = [g.next_field_element_without_rejection() for _ in range(T)]
    assert len(set(xs + ys)) == 2 * T
    mds = [[pow((xs[i] + ys[j]) % R, -1, R) for j in range(T)] for i in range(T)]
    return rc, mds


_CONST = None


def constants():
    global _CONST
    if _CONST is None:
        _CONST = generate_constants()
    return _CONST


def permute(state):
    """Hades permutation x5, t = 3: R_F/2 full rounds, R_P partial rounds (S-box on word 0), R_F/2 full rounds."""
    rc, mds = constants()
    s = list(state)
    half = R_F // 2
    for r in range(R_F + R_P):
        s = [(a + c) % R for a, c in zip(s, rc[r])]
        if r < half or r >= half + R_P:
            s = [pow(a, ALPHA, R) for a in s]
        else:
            s[0] = pow(s[0], ALPHA, R)
        s = [sum(mds[i][j] * s[j] for j in range(T)) % R for i in range(T)]
    return s


class Sponge:
    """snark-verifier `Poseidon<F, L, T, RATE>` (util/hash/poseidon.rs) with the native loader."""

    def __init__(self):
        self.state = [1 << 64, 0, 0]
        self.buf = []
        self.permutations = 0

    def update(self, elements):
        self.buf.extend(e % R for e in elements)

    def _absorb(self, chunk):
        # absorb_with_pre_constants: inputs added to words 1.., then a single 1 in the first free word (if any)
        s = self.state
        for i, v in enumerate(chunk):
            s[1 + i] = (s[1 + i] + v) % R
        if len(chunk) < RATE:
            s[1 + len(chunk)] = (s[1 + len(chunk)] + 1) % R
        self.state = permute(s)
        self.permutations += 1

    def absorb_full_chunks(self):
        """Runs the permutations of the FULL chunks buffered so far and keeps the remainder (< RATE values) buffered: what squeeze()
        would do with them first, whatever is absorbed later -- a full chunk is absorbed without padding and the "exact multiple"
        rule only looks at the tail.  Lets a caller remember the state behind a long common prefix (halo2_ref._absorb_public_inputs)."""
        full = len(self.buf) - len(self.buf) % RATE
        for i in range(0, full, RATE):
            self._absorb(self.buf[i:i + RATE])
        self.buf = self.buf[full:]

    def squeeze(self):
        buf, self.buf = self.buf, []
        exact = len(buf) % RATE == 0
        for i in range(0, len(buf), RATE):
            self._absorb(buf[i:i + RATE])
        if exact:
            self._absorb([])
        return self.state[1]


from oracle.point_encoding import point_compress, point_decompress  # noqa: E402,F401  (the one definition of the layout)


class PoseidonTranscript:
    """snark-verifier `PoseidonTranscript<G1Affine, NativeLoader, S, 3, 2, 8, 57>` as both writer and reader."""

    def __init__(self, proof=None):
        self.sp = Sponge()
        self.out = bytearray()
        self.inp = proof
        self.pos = 0

    def common_point(self, P):
        if P is None:
            raise ValueError("Cannot write points at infinity to the transcript")
        self.sp.update([P[0] % R, P[1] % R])   # fe_to_fe::<Fq, Fr>: the integer value reduced mod r

    def common_scalar(self, s):
        self.sp.update([s % R])

    def write_point(self, P):
        self.common_point(P)
        self.out += point_compress(P)

    def write_scalar(self, s):
        self.common_scalar(s)
        self.out += (s % R).to_bytes(32, "little")

    def read_point(self):
        assert self.pos + 32 <= len(self.inp), "proof too short"
        P = point_decompress(self.inp[self.pos:self.pos + 32])
        self.pos += 32
        self.common_point(P)
        return P

    def read_scalar(self):
        assert self.pos + 32 <= len(self.inp), "proof too short"
        s = int.from_bytes(self.inp[self.pos:self.pos + 32], "little")
        assert s < R, "scalar not canonical"
        self.pos += 32
        self.common_scalar(s)
        return s

    def squeeze(self):
        return self.sp.squeeze()
