#!/usr/bin/env python3
"""Golden digests of the ORACLE prover's proofs for sampled members of BASELINE configs[2]'s batch of 64.

tests/test_batch_gloo.py proves the 64 inputs of zk_fhe_amd.inputs.config3_batch (the reference's bfv.in + 63 seeded vectors) with the
seeds b"batch64-<i>" on the GPU.  This script makes the proofs of a few of them with the Python oracle (oracle/halo2_ref.py: no line shared
with the product; ~40 s each at k = 13) and commits their SHA-256: the batch is then checked against an independent prover, not only
against the native CPU prover (which shares the product's host code).

    python tests/golden/gen_batch64_digests.py        # writes tests/golden/batch64_proofs.json
"""
import hashlib
import importlib.util
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import circuit_ref as C  # noqa: E402
from oracle import halo2_ref as H  # noqa: E402

SAMPLES = (0, 13, 38, 63)


def main():
    spec = importlib.util.spec_from_file_location("zkfhe_inputs", os.path.join(ROOT, "zk-fhe_amd", "inputs.py"))   # no need for the .so
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    G = os.path.join(ROOT, "tests", "golden", "bfv")
    cfgj = json.load(open(os.path.join(G, "bfv_config.json")))
    hcfg = H.Config.from_pinning(cfgj)
    bp = {"gate0": cfgj["break_points"]["gate"][0], "gate1": cfgj["break_points"]["gate"][1], "rlc": cfgj["break_points"]["rlc"]}
    prm = C.BfvParams()
    srs = H.make_srs(13)
    pk, _ = H.keygen_circuit(hcfg, H.BfvCircuit(json.load(open(os.path.join(G, "bfv_empty.in"))), prm), srs, bp)
    inputs = gen.config3_batch(open(os.path.join(G, "bfv.in"), "rb").read(), 64)
    out = {"seed_format": "batch64-%d", "count": 64, "proofs": {}}
    for i in SAMPLES:
        proof, inst = H.prove(hcfg, pk, srs, H.BfvCircuit(json.loads(inputs[i]), prm), b"batch64-%d" % i)
        assert H.verify(H.VerifyingKey(pk), srs, inst, proof)
        out["proofs"][str(i)] = {"proof_len": len(proof), "proof_sha256": hashlib.sha256(proof).hexdigest(),
                                 "input_sha256": hashlib.sha256(inputs[i]).hexdigest()}
        print(i, out["proofs"][str(i)]["proof_sha256"], flush=True)
    json.dump(out, open(os.path.join(HERE, "batch64_proofs.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
