"""Steady-state prove timings for the larger configurations (BASELINE configs 4 and 5): python tools/exp/seq_configs.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import zk_fhe_amd as zk  # noqa: E402
from zk_fhe_amd import inputs  # noqa: E402

ctx = zk.Context(0)


def run(N, k, Q, reps):
    text = json.dumps(inputs.generate(N, Q, 7, 19, seed=3))
    probe = zk.bfv_build_tables(text, (N, Q, 7, 19), zk.BfvConfig(k, 8, 400, 120, 16, 109), 1, keygen_mode=False)
    n0, n1, nr = (len(probe["break_points"][w]) + 1 for w in ("gate0", "gate1", "rlc"))
    nl = -(-probe["lookups"] // ((1 << k) - 109))
    srs = zk.Srs(ctx, k)
    pk = zk.BfvProvingKey(ctx, srs, text, (N, Q, 7, 19), zk.BfvConfig(k, n0, n1, nl, nr, 109))
    for r in range(reps):
        proof, inst, tm = pk.prove(text, b"s%d" % r)
        print("k=%d N=%d columns %s rep %d [witness, commit, quotient, open, total] ms: %s" % (k, N, (n0, n1, nl, nr), r, [round(x, 1) for x in tm]), flush=True)
    pk.destroy()
    srs.destroy()


run(1024, 13, 536870909, 2)
run(4096, 16, (1 << 60) - 93, 3)
run(16384, 19, (1 << 60) - 93, 3)
