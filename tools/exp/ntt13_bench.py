#!/usr/bin/env python3
"""Times the 2^13 tile: plain forward / inverse NTT and the 3-row coset extension, 256 columns (GPU box)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    import torch  # noqa: F401
    import zk_fhe_amd as zk
    from oracle import binding as orc
    ctx = zk.Context(0)
    rng = np.random.default_rng(1)
    n_cols, log_n = 256, 13
    n = 1 << log_n
    raw = np.frombuffer(rng.bytes(32 * n * n_cols), dtype=np.uint64).reshape(n_cols, n, 4).copy()
    raw[..., 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
    # correctness on 3 columns first
    for inv in (False, True):
        got = ctx.ntt(raw[:3], log_n, inverse=inv)
        want = orc.ntt(raw[:3], log_n, inv)
        print("ntt inverse=%s matches oracle:" % inv, bool(np.array_equal(got, want)), flush=True)
    d = ctx.to_device(raw)
    d2 = ctx.alloc(n_cols * n * 32)
    for inv in (False, True):
        for _ in range(3):
            ctx.ntt_to_dev(d, d2, n_cols, log_n, inverse=inv)
        ts = []
        for _ in range(20):
            ctx.timer_start()
            ctx.ntt_to_dev(d, d2, n_cols, log_n, inverse=inv)
            ts.append(ctx.timer_stop_ms())
        print("2^13 x %d inverse=%s out of place: median %.4f ms, min %.4f ms" % (n_cols, inv, float(np.median(ts)), min(ts)), flush=True)
    ts = []
    for _ in range(10):
        ctx.timer_start()
        ctx.ntt_dev(d, n_cols, log_n, inverse=False)
        ts.append(ctx.timer_stop_ms())
    print("2^13 x %d in place (through scratch): median %.4f ms" % (n_cols, float(np.median(ts))), flush=True)
    g = orc.ints_to_mont([7])[0]
    o = ctx.alloc(n_cols * n * 4 * 32)
    for _ in range(3):
        ctx.coset_ntt_dev(d, o, n_cols, log_n, 2, g)
    ts = []
    for _ in range(20):
        ctx.timer_start()
        ctx.coset_ntt_dev(d, o, n_cols, log_n, 2, g)
        ts.append(ctx.timer_stop_ms())
    print("coset 2^13 x %d x 4 rows: median %.4f ms (%.4f per 256 tiles)" % (n_cols, float(np.median(ts)), float(np.median(ts)) / 4), flush=True)


if __name__ == "__main__":
    main()
