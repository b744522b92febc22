"""The C++ verifier behind zkfhe_bfv_verify (host CPU, pairing check) against proofs made by the oracle prover.
CPU only.  (GPU-made proofs are checked by both verifiers in tests/test_gpu_prover.py.)"""
import pytest

import zk_fhe_amd as zk
from oracle import circuit_ref as C
from oracle import halo2_ref as H
from tests.test_proof_oracle import synth_input


def make(transcript):
    prm = C.BfvParams(N=8)
    inp = synth_input(8, prm.Q, prm.T, prm.B, 1)
    circ = H.BfvCircuit(inp, prm)
    cfg = H.auto_config(9, 9, circ, transcript=transcript)
    srs = H.make_srs(9)
    pk, _ = H.keygen_circuit(cfg, circ, srs)
    proof, inst = H.prove(cfg, pk, srs, circ, b"seed-v")
    vkb = zk.make_vk_bytes(cfg.k, cfg.n_gate0, cfg.n_gate1, cfg.n_lookup, cfg.n_rlc, cfg.unusable_rows, cfg.lookup_bits,
                           pk.vk_digest, pk.fixed_commit, pk.sigma_commit, transcript)
    return vkb, inst, proof


@pytest.fixture(scope="module")
def toy():
    return make("poseidon")   # the reference's transcript (snark-verifier PoseidonTranscript)


def test_accepts_oracle_proof_blake2b_transcript():
    vkb, inst, proof = make("blake2b")
    ok, why = zk.bfv_verify(vkb, inst, proof)
    assert ok, why
    bad = bytearray(proof)
    bad[len(proof) // 2] ^= 1
    assert not zk.bfv_verify(vkb, inst, bytes(bad))[0]
    # a proof made for one transcript does not verify under the other: the kind is part of the verifying key
    vkp, instp, proofp = make("poseidon")
    assert not zk.bfv_verify(vkb, instp, proofp)[0] and not zk.bfv_verify(vkp, inst, proof)[0]


def test_accepts_oracle_proof(toy):
    vkb, inst, proof = toy
    ok, why = zk.bfv_verify(vkb, inst, proof)
    assert ok, why


def test_rejects_tampering(toy):
    vkb, inst, proof = toy
    for pos in (40, len(proof) // 2, len(proof) - 40, len(proof) - 1):
        bad = bytearray(proof)
        bad[pos] ^= 1
        ok, _ = zk.bfv_verify(vkb, inst, bytes(bad))
        assert not ok
    inst2 = list(inst)
    inst2[0] = (inst2[0] + 1) % H.R
    assert not zk.bfv_verify(vkb, inst2, proof)[0]
    assert not zk.bfv_verify(vkb, inst, proof[:-32])[0]
    assert not zk.bfv_verify(vkb, inst, proof, srs_seed=b"another-srs")[0]
    vk2 = bytearray(vkb)
    vk2[100] ^= 1
    ok, why = zk.bfv_verify(bytes(vk2), inst, proof)
    assert not ok and "digest" in why     # the vk digest is recomputed from the file's contents, not trusted
    # halo2 Error::InstanceTooLarge: L_{i+n}(x) = L_i(x), so a longer instance vector could alias the committed one
    n, u = 1 << 9, (1 << 9) - (9 - 3) - 1
    moved = list(inst) + [0] * (n + 4 - len(inst))
    moved[3], moved[n + 3] = (moved[3] - 5) % H.R, 5
    ok, why = zk.bfv_verify(vkb, moved, proof)
    assert not ok and "instances" in why
    assert len(inst) <= u


def test_external_srs_verifier_half(toy):
    """zkfhe_bfv_verify_g2: the same check with G2 and s*G2 passed in (an external ceremony) instead of derived from a seed."""
    from oracle import pairing_ref as PR
    vkb, inst, proof = toy
    srs = H.srs_verifier_half(9)
    pt = lambda P: ((P[0].c[0], P[0].c[1]), (P[1].c[0], P[1].c[1]))  # noqa: E731
    g2, s_g2 = pt(PR.G2_GEN), pt(srs["s_g2"])
    ok, why = zk.bfv_verify(vkb, inst, proof, g2=g2, s_g2=s_g2)
    assert ok, why
    other = pt(PR.ec_mul(PR.G2_GEN, srs["s"] + 1))
    assert not zk.bfv_verify(vkb, inst, proof, g2=g2, s_g2=other)[0]
    off_curve = ((s_g2[0][0] + 1, s_g2[0][1]), s_g2[1])
    ok, why = zk.bfv_verify(vkb, inst, proof, g2=g2, s_g2=off_curve)
    assert not ok and "curve" in why
