"""The native CPU prover behind bench.py's cpu_baseline leg (oracle/cpu_prover.cpp) against the oracle prover
(oracle/halo2_ref.py): same key, same input, same seed => the same proof bytes, and the oracle's verifier accepts them.
Both are test infrastructure; neither is reachable from the product."""
import json

import pytest

from oracle import circuit_ref as C
from oracle import cpu_prover as CP
from oracle import halo2_ref as H
from tests.test_proof_oracle import synth_input


@pytest.fixture(scope="module")
def toy():
    prm = C.BfvParams(N=8)
    inp = synth_input(8, prm.Q, prm.T, prm.B, 1)
    return prm, inp, H.make_srs(9)


@pytest.mark.parametrize("transcript", ["poseidon", "blake2b"])
def test_cpu_prover_bytes_equal_oracle_prover(toy, transcript):
    prm, inp, srs = toy
    circ = H.BfvCircuit(inp, prm)
    cfg = H.auto_config(9, 9, circ, transcript=transcript)
    pk, _ = H.keygen_circuit(cfg, circ, srs)
    cp = CP.CpuProver(cfg, pk, srs, prm)
    for seed in (b"seed-1", b"another seed that is longer than thirty-two bytes, so it is hashed"):
        proof_o, inst = H.prove(cfg, pk, srs, H.BfvCircuit(inp, prm), seed)
        proof_c = cp.prove(json.dumps(inp), seed)
        assert proof_c == proof_o
        assert H.verify(H.VerifyingKey(pk), srs, inst, proof_c)
    assert set(cp.phase_ms) == set(CP.PHASES) and all(v >= 0 for v in cp.phase_ms.values())
    # a second encryption under the same key: the prover object is reusable
    inp2 = synth_input(8, prm.Q, prm.T, prm.B, 2)
    proof_o2, _ = H.prove(cfg, pk, srs, H.BfvCircuit(inp2, prm), b"s2")
    assert cp.prove(json.dumps(inp2), b"s2") == proof_o2
    cp.close()


def test_cpu_prover_rejects_a_wrong_ciphertext(toy):
    prm, inp, srs = toy
    circ = H.BfvCircuit(inp, prm)
    cfg = H.auto_config(9, 9, circ)
    pk, _ = H.keygen_circuit(cfg, circ, srs)
    cp = CP.CpuProver(cfg, pk, srs, prm)
    bad = dict(inp)
    bad["c0"] = list(inp["c0"])
    bad["c0"][0] = str((int(bad["c0"][0]) + 1) % prm.Q)
    with pytest.raises(RuntimeError, match="does not close|constraint is violated|not in table"):
        cp.prove(json.dumps(bad), b"seed-1")
    cp.close()
