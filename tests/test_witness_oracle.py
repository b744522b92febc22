"""Pins the witness-generation oracle (oracle/circuit_ref.py) against the reference's own data:
  KAT 1  data/bfv/bfv.in ciphertext identity            
  KAT 2  intermediate witnesses derived by following src/poly.rs literally
  KAT 3  configs/bfv.json: column counts and all 158 break points of the pinned layout
Fixtures under tests/golden/bfv/ are verbatim copies of the reference's DATA files."""
import json
import os

import pytest

from oracle import circuit_ref as C

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden", "bfv")


@pytest.fixture(scope="module")
def bfv():
    inp = C.load_input(os.path.join(G, "bfv.in"))
    prm = C.BfvParams()
    ctx0, pub, st = C.bfv_phase0(inp, prm)
    gamma = 0x1234567890ABCDEF1234567890ABCDEF  # any challenge: the layout does not depend on it
    ctx_gate, ctx_rlc = C.bfv_phase1(st, prm, gamma)
    return inp, prm, ctx0, pub, st, ctx_gate, ctx_rlc, gamma


def test_kat1_ciphertext(bfv):
    inp, prm = bfv[0], bfv[1]
    N, Q, T = prm.N, prm.Q, prm.T
    pk0, pk1, m, u, e0, e1, c0, c1 = ([int(x) for x in inp[k]] for k in ("pk0", "pk1", "m", "u", "e0", "e1", "c0", "c1"))
    assert [int(x) for x in inp["cyclo"]] == [1] + [0] * (N - 1) + [1]

    def negacyclic(a, b):  # big-endian coefficient order
        a, b = a[::-1], b[::-1]
        out = [0] * N
        for i in range(N):
            if b[i] == 0:
                continue
            for j in range(N):
                k = i + j
                if k < N:
                    out[k] += a[j] * b[i]
                else:
                    out[k - N] -= a[j] * b[i]
        return [x % Q for x in out][::-1]
    pu0, pu1 = negacyclic(pk0, u), negacyclic(pk1, u)
    assert [(pu0[i] + (Q // T) * m[i] + e0[i]) % Q for i in range(N)] == c0
    assert [(pu1[i] + e1[i]) % Q for i in range(N)] == c1


def test_kat2_intermediates(bfv):
    st = bfv[4]
    un = st["unassigned"]
    Q = 536870909
    assert len(un["pk0_u"].coefficients) == 2047
    assert un["pk0_u"].max_bits == 68
    assert max(c.bit_length() for c in un["pk0_u"].coefficients) == 66
    assert un["pk0_u"].coefficients[0] == 162695937477549308 and un["pk0_u"].coefficients[-1] == 28861834
    assert un["pk1_u"].coefficients[0] == 10207660791531000 and un["pk1_u"].coefficients[-1] == 214054897
    q0 = un["quotient_0"].coefficients
    assert len(q0) == 1025 and q0[:2] == [0, 0] and q0[2] == 233826108
    q1 = un["quotient_1"].coefficients
    assert next(c for c in q1 if c) == 517857659
    r0 = un["remainder_0"].coefficients
    assert len(r0) == 2049 and all(c == 0 for c in r0[:1025])
    assert len(un["q0c"].coefficients) == 2049 and max(c.bit_length() for c in un["q0c"].coefficients) <= 29
    lhs = [(a + b) % Q for a, b in zip(un["q0c"].coefficients, r0)]
    assert lhs == [0, 0] + [c % Q for c in un["pk0_u"].coefficients]


def test_kat3_layout(bfv):
    inp, prm, ctx0, pub, st, ctx_gate, ctx_rlc, gamma = bfv
    cfg = json.load(open(os.path.join(G, "bfv_config.json")))
    p = cfg["params"]
    k, unusable = p["degree"], p["unusable_rows"]
    max_rows = (1 << k) - unusable
    assert max_rows == 8083
    assert len(pub) == 5121
    assert len(ctx0.advice) == 23558
    assert len(ctx_gate.advice) == 1231992
    assert len(ctx_rlc.advice) == 32764
    assert len(ctx_gate.lookup) == 286756 and not ctx0.lookup
    _, _, bp0, ncol0 = C.place_stream(len(ctx0.advice), ctx0.selector, max_rows)
    _, _, bp1, ncol1 = C.place_stream(len(ctx_gate.advice), ctx_gate.selector, max_rows)
    _, _, bpr, ncolr = C.place_stream(len(ctx_rlc.advice), ctx_rlc.selector, max_rows, rlc=True)
    assert [ncol0, ncol1, 0] == p["num_range_advice"]
    assert ncolr == p["num_rlc_columns"]
    assert [0, -(-len(ctx_gate.lookup) // max_rows), 0] == p["num_lookup_advice"]
    assert bp0 == cfg["break_points"]["gate"][0]
    assert bp1 == cfg["break_points"]["gate"][1]
    assert bpr == cfg["break_points"]["rlc"]
    assert len(bp0) + len(bp1) + len(bpr) == 158
    # replaying the pinned break points (prover stage) gives the same placement
    pl_a = C.place_stream(len(ctx_gate.advice), ctx_gate.selector, max_rows)[0]
    pl_b = C.place_stream(len(ctx_gate.advice), ctx_gate.selector, max_rows, break_points=cfg["break_points"]["gate"][1])[0]
    assert pl_a == pl_b


def test_gates_hold(bfv):
    """mock-prover style check on the raw streams: q*(a + b*c - d) = 0 and RLC q*(a*gamma + b - c) = 0."""
    inp, prm, ctx0, pub, st, ctx_gate, ctx_rlc, gamma = bfv
    R = C.R
    for ctx in (ctx0, ctx_gate):
        a = ctx.advice
        for o in ctx.selector:
            assert (a[o] + a[o + 1] * a[o + 2] - a[o + 3]) % R == 0
    a = ctx_rlc.advice
    for o in ctx_rlc.selector:
        assert (a[o] * gamma + a[o + 1] - a[o + 2]) % R == 0
    vals = {}
    for ctx in (ctx0, ctx_gate, ctx_rlc):
        vals[ctx.cid] = ctx.advice
    for ctx in (ctx0, ctx_gate, ctx_rlc):
        for (c1, o1), (c2, o2) in ctx.copies:
            assert vals[c1][o1] == vals[c2][o2]
        for (c1, o1), v in ctx.consts:
            assert vals[c1][o1] == v
        for (c1, o1) in ctx.lookup:
            assert 0 <= vals[c1][o1] < 256


def test_empty_input_keygen_branch():
    """bfv_empty.in: all-zero witnesses take the zero branch of divide_by_cyclo (src/poly.rs:118-123)."""
    inp = C.load_input(os.path.join(G, "bfv_empty.in"))
    prm = C.BfvParams()
    ctx0, pub, st = C.bfv_phase0(inp, prm)
    assert len(ctx0.advice) == 23558
    assert st["quotient_0"].max_num_bits == 29 and st["remainder_0"].degree == 2048
    ctx_gate, ctx_rlc = C.bfv_phase1(st, prm, 5)
    assert len(ctx_gate.advice) == 1231992 and len(ctx_rlc.advice) == 32764
