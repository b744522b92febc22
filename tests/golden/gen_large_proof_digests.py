#!/usr/bin/env python3
"""Golden digests of the ORACLE prover's proofs at the large configurations (BASELINE configs[3] / [4]).

The Python oracle (oracle/halo2_ref.py + oracle/circuit_ref.py: no line shared with the product) takes minutes to hours at these sizes, so
the GPU suite cannot run it; this script runs it once, here, on the CPU, and commits what it produced: the SHA-256 of the proof bytes and of
the public inputs, the column counts of the oracle's own auto-configuration and the verifying-key digest of the oracle's keygen, for the
input of tests/large_inputs.py and the seed the GPU tests prove with.  tests/test_gpu_prover.py then requires the GPU proof of the same
input and seed to have exactly these digests: byte parity at k = 16 (and k = 19) against a prover that shares nothing with the product.

    python tests/golden/gen_large_proof_digests.py k16 [k19]        # writes / updates tests/golden/large_proofs.json
"""
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import circuit_ref as C  # noqa: E402
from oracle import halo2_ref as H  # noqa: E402
from tests.large_inputs import B, Q60, T, large_input  # noqa: E402

CONFIGS = {"k16": (4096, 16, "config4"), "k19": (16384, 19, "config5")}


def main():
    path = os.path.join(HERE, "large_proofs.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    for name in sys.argv[1:]:
        N, k, tag = CONFIGS[name]
        t0 = time.time()
        inp = large_input(N)
        prm = C.BfvParams(N=N, Q=Q60, T=T, B=B)
        circ = H.BfvCircuit(inp, prm)
        hcfg = H.auto_config(k, 109, circ)
        print(name, "columns", hcfg.n_gate0, hcfg.n_gate1, hcfg.n_lookup, hcfg.n_rlc, "%.0f s" % (time.time() - t0), flush=True)
        srs = H.make_srs(k)
        print(name, "srs %.0f s" % (time.time() - t0), flush=True)
        pk, _ = H.keygen_circuit(hcfg, circ, srs)
        print(name, "keygen %.0f s" % (time.time() - t0), flush=True)
        proof, inst = H.prove(hcfg, pk, srs, circ, tag.encode())
        assert H.verify(H.VerifyingKey(pk), srs, inst, proof)
        out[name] = {"N": N, "k": k, "Q": Q60, "seed": tag, "transcript": "poseidon", "unusable_rows": 109,
                     "columns": [hcfg.n_gate0, hcfg.n_gate1, hcfg.n_lookup, hcfg.n_rlc],
                     "vk_digest": "%064x" % pk.vk_digest, "proof_len": len(proof), "proof_sha256": hashlib.sha256(proof).hexdigest(),
                     "proof_head_hex": proof[:64].hex(), "proof_tail_hex": proof[-64:].hex(),
                     "instances": len(inst), "instances_sha256": hashlib.sha256(b"".join(int(v).to_bytes(32, "little") for v in inst)).hexdigest(),
                     "oracle_seconds": round(time.time() - t0)}
        json.dump(out, open(path, "w"), indent=1, sort_keys=True)
        print(name, "done in %.0f s" % (time.time() - t0), out[name]["proof_sha256"], flush=True)


if __name__ == "__main__":
    main()
