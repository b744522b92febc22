import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import zk_fhe_amd as zk
ctx = zk.Context(0)
for cols in (256, 1624):
    buf = ctx.alloc(cols * (32 << 13))
    ctx._check(ctx.lib.zkfhe_memset_dev(ctx.h, buf.at(0), 1, buf.nbytes))
    for _ in range(3): ctx.ntt_dev(buf, cols, 13, False)
    ts = []
    for _ in range(10):
        ctx.timer_start(); ctx.ntt_dev(buf, cols, 13, False); ts.append(ctx.timer_stop_ms())
    print("ntt 2^13 x %d: %.3f ms  (%.2f us per column)" % (cols, np.median(ts), 1e3 * np.median(ts) / cols))
