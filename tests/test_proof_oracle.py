"""The oracle proof system (oracle/halo2_ref.py) end to end on the CPU: keygen -> prove -> verify, with a real
BN254 pairing check at the end.  A small BFV instance (N = 8) keeps the default run short; the full
reference vector (data/bfv/bfv.in, k = 13, pinned configs/bfv.json layout) runs as well (~40 s, 8 cores)."""
import json
import os
import random

import pytest

from oracle import circuit_ref as C
from oracle import halo2_ref as H

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden", "bfv")


def synth_input(N, Q, T, B, seed):
    """BFV encryption of a random message, formula checked against bfv.in in test_witness_oracle (KAT 1)."""
    rng = random.Random(seed)
    pk0 = [rng.randrange(Q) for _ in range(N)]
    pk1 = [rng.randrange(Q) for _ in range(N)]
    u = [rng.choice([0, 1, Q - 1]) for _ in range(N)]
    m = [rng.choice(list(range(0, T // 2 + 1)) + [Q - i for i in range(1, T // 2 + 1)]) for _ in range(N)]
    e0 = [rng.choice(list(range(0, B + 1)) + [Q - i for i in range(1, B + 1)]) for _ in range(N)]
    e1 = [rng.choice(list(range(0, B + 1)) + [Q - i for i in range(1, B + 1)]) for _ in range(N)]

    def nega(a, b):
        a, b = a[::-1], b[::-1]
        out = [0] * N
        for i in range(N):
            for j in range(N):
                k = i + j
                if k < N:
                    out[k] += a[j] * b[i]
                else:
                    out[k - N] -= a[j] * b[i]
        return [x % Q for x in out][::-1]
    pu0, pu1 = nega(pk0, u), nega(pk1, u)
    c0 = [(pu0[i] + (Q // T) * m[i] + e0[i]) % Q for i in range(N)]
    c1 = [(pu1[i] + e1[i]) % Q for i in range(N)]
    s = lambda v: [str(x) for x in v]  # noqa: E731
    return dict(pk0=s(pk0), pk1=s(pk1), m=s(m), u=s(u), e0=s(e0), e1=s(e1), c0=s(c0), c1=s(c1), cyclo=s([1] + [0] * (N - 1) + [1]))


@pytest.fixture(scope="module")
def toy():
    prm = C.BfvParams(N=8)
    inp = synth_input(8, prm.Q, prm.T, prm.B, 1)
    circ = H.BfvCircuit(inp, prm)
    cfg = H.auto_config(9, 9, circ)
    srs = H.make_srs(9)
    pk, _ = H.keygen_circuit(cfg, circ, srs)
    return prm, inp, circ, cfg, srs, pk


def test_toy_prove_verify(toy):
    prm, inp, circ, cfg, srs, pk = toy
    proof, inst = H.prove(cfg, pk, srs, circ, b"seed-1")
    vk = H.VerifyingKey(pk)
    assert H.verify(vk, srs, inst, proof)
    proof2, _ = H.prove(cfg, pk, srs, circ, b"seed-1")
    assert proof2 == proof, "same seed must give the same bytes"
    assert H.prove(cfg, pk, srs, circ, b"seed-2")[0] != proof
    for pos in (0, len(proof) // 2, len(proof) - 1):   # a commitment, an evaluation, the last opening point
        bad = bytearray(proof)
        bad[pos] ^= 1
        assert not H.verify(vk, srs, inst, bytes(bad))
    assert not H.verify(vk, srs, inst, proof[:-32]) and not H.verify(vk, srs, inst, proof + bytes(32))
    inst2 = list(inst)
    inst2[3] = (inst2[3] + 1) % H.R
    assert not H.verify(vk, srs, inst2, proof)


def test_toy_wrong_witness_is_rejected(toy):
    prm, inp, circ, cfg, srs, pk = toy
    bad = dict(inp)
    c0 = list(bad["c0"])
    c0[2] = str((int(c0[2]) + 1) % prm.Q)
    bad["c0"] = c0
    with pytest.raises(AssertionError):
        H.prove(cfg, pk, srs, H.BfvCircuit(bad, prm), b"seed-1")


def test_bfv_in_k13_prove_verify():
    cfgj = json.load(open(os.path.join(G, "bfv_config.json")))
    cfg = H.Config.from_pinning(cfgj)
    bp = {"gate0": cfgj["break_points"]["gate"][0], "gate1": cfgj["break_points"]["gate"][1], "rlc": cfgj["break_points"]["rlc"]}
    prm = C.BfvParams()
    srs = H.make_srs(13)
    pk, _ = H.keygen_circuit(cfg, H.BfvCircuit(C.load_input(os.path.join(G, "bfv_empty.in")), prm), srs, bp)
    proof, inst = H.prove(cfg, pk, srs, H.BfvCircuit(C.load_input(os.path.join(G, "bfv.in")), prm), b"seed-1")
    assert len(inst) == 5121
    assert H.verify(H.VerifyingKey(pk), srs, inst, proof)


def test_verifier_remembers_public_input_state_without_changing_answers(toy):
    """halo2_ref._absorb_public_inputs keeps the hash state behind `vk digest | public inputs` (the large-configuration tests verify
    one proof and its damaged copies against the same 81 921 inputs): a second verification must give the same answers as the
    first, a changed public input must miss, and Sponge.absorb_full_chunks must not change what a sponge squeezes."""
    from oracle import poseidon_ref as P
    prm, inp, circ, cfg, srs, pk = toy
    vk = H.VerifyingKey(pk)
    for tcfg in (cfg, H.Config(cfg.k, cfg.n_gate0, cfg.n_gate1, cfg.n_lookup, cfg.n_rlc, cfg.unusable_rows, transcript="blake2b")):
        pk_t = pk if tcfg is cfg else H.keygen_circuit(tcfg, circ, srs)[0]
        vk = H.VerifyingKey(pk_t)
        proof, inst = H.prove(tcfg, pk_t, srs, circ, b"memo")
        H._PUBLIC_INPUT_STATES.clear()
        assert H.verify(vk, srs, inst, proof) and len(H._PUBLIC_INPUT_STATES) == 1      # miss: absorbed and stored
        assert H.verify(vk, srs, inst, proof) and len(H._PUBLIC_INPUT_STATES) == 1      # hit: same verdict
        bad = bytearray(proof)
        bad[len(proof) // 2] ^= 1
        assert not H.verify(vk, srs, inst, bytes(bad))
        inst2 = list(inst)
        inst2[1] = (inst2[1] + 1) % H.R
        assert not H.verify(vk, srs, inst2, proof) and len(H._PUBLIC_INPUT_STATES) == 2  # other public inputs: their own entry
        assert H.verify(vk, srs, inst, proof)
    for n in range(0, 7):
        for cut in range(0, n + 1):
            a, b = P.Sponge(), P.Sponge()
            vals = [3 * i + 1 for i in range(n)]
            a.update(vals)
            b.update(vals[:cut])
            b.absorb_full_chunks()
            b.update(vals[cut:])
            assert a.squeeze() == b.squeeze() and a.squeeze() == b.squeeze(), (n, cut)
