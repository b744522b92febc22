"""zk_fhe_amd.inputs: the BFV input generator (the in-tree replacement of the external bfv-py, reference README.md:25).
Checks: the ciphertext decrypts to the message; the output has the shape of data/bfv/bfv.in; the oracle's restatement
of the circuit accepts it (every gadget assertion of examples/bfv.rs holds), for the reference's parameters and a 60-bit Q."""
import importlib
import json
import os

import numpy as np
import pytest

from oracle import circuit_ref as C

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def inputs():
    # the module has no dependency on the HIP library; import it by path so the test runs without the .so
    spec = importlib.util.spec_from_file_location("zkfhe_inputs", os.path.join(os.path.dirname(HERE), "zk-fhe_amd", "inputs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("n,q", [(64, 536870909), (32, (1 << 60) - 93)])
def test_generated_ciphertext_decrypts_and_satisfies_the_circuit(inputs, n, q):
    t, b = 7, 19
    inp, sec = inputs.generate(n, q, t, b, seed=11, with_secret=True)
    assert (inputs.decrypt(sec["sk"], sec["c0"], sec["c1"], q, t) == sec["m"]).all()
    ref = json.load(open(os.path.join(HERE, "golden", "bfv", "bfv.in")))
    assert set(inp) == set(ref)
    assert all(len(inp[k]) == (n + 1 if k == "cyclo" else n) for k in inp)
    assert all(0 <= int(x) < q for k in inp for x in inp[k])
    # mock-prover check of the oracle's restatement of the circuit: gates, copies, constants, lookups all hold
    prm = C.BfvParams(N=n, Q=q, T=t, B=b)
    ctx0, pub, st = C.bfv_phase0(inp, prm)
    gamma = 0x1234567890ABCDEF1234567890ABCDEF
    ctx_gate, ctx_rlc = C.bfv_phase1(st, prm, gamma)
    R = C.R
    for ctx in (ctx0, ctx_gate):
        a = ctx.advice
        assert all((a[o] + a[o + 1] * a[o + 2] - a[o + 3]) % R == 0 for o in ctx.selector)
    a = ctx_rlc.advice
    assert all((a[o] * gamma + a[o + 1] - a[o + 2]) % R == 0 for o in ctx_rlc.selector)
    vals = {ctx.cid: ctx.advice for ctx in (ctx0, ctx_gate, ctx_rlc)}
    for ctx in (ctx0, ctx_gate, ctx_rlc):
        assert all(vals[c1][o1] == vals[c2][o2] for (c1, o1), (c2, o2) in ctx.copies)
        assert all(vals[c1][o1] == v for (c1, o1), v in ctx.consts)
        assert all(0 <= vals[c1][o1] < 256 for (c1, o1) in ctx.lookup)


def test_generator_is_deterministic_and_seed_sensitive(inputs):
    a = inputs.generate(16, 536870909, 7, 19, seed=3)
    assert a == inputs.generate(16, 536870909, 7, 19, seed=3)
    assert a != inputs.generate(16, 536870909, 7, 19, seed=4)
    e = inputs.empty(16)
    assert len(e["cyclo"]) == 17 and set(e["pk0"]) == {"0"}


def test_one_key_many_encryptions_and_config3_batch(inputs):
    """key_seed fixes the key pair: different seeds encrypt different messages under one public key, and each still decrypts.
    config3_batch = SURVEY.md 8(d) config 3: the reference's bfv.in first, then seeded vectors that are valid encryptions
    (c0 = pk0 u + floor(Q/T) m + e0, c1 = pk1 u + e1) with a public key of their own each."""
    q, t, b = 536870909, 7, 19
    (a, sa), (c, sc) = (inputs.generate(32, q, t, b, seed=s, with_secret=True, key_seed=77) for s in (1, 2))
    assert a["pk0"] == c["pk0"] and a["pk1"] == c["pk1"] and a["c0"] != c["c0"] and (sa["sk"] == sc["sk"]).all()
    for s in (sa, sc):
        assert (inputs.decrypt(s["sk"], s["c0"], s["c1"], q, t) == s["m"]).all()
    ref = open(os.path.join(HERE, "golden", "bfv", "bfv.in"), "rb").read()
    batch = inputs.config3_batch(ref, 5)
    assert batch[0] == ref and len(batch) == 5 and len({x for x in batch}) == 5
    prm = C.BfvParams()
    for text in batch[1:3]:
        inp = json.loads(text)
        assert inp["pk0"] != json.loads(batch[0])["pk0"]
        C.bfv_phase0(inp, prm)      # asserts the ciphertext identity (division by the cyclotomic polynomial leaves the stated remainder)
    assert batch[1] == inputs.config3_vector(20240613 + 1).encode()
