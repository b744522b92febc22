#!/usr/bin/env python3
"""GPU occupancy of the tail of a rocprofv3 kernel trace in time bins: fraction of each bin in which at least one kernel ran,
average number of kernels executing, and the kernel with the largest share.  usage: tools/busy_bins.py <results.db> [tail_ms] [bin_ms]"""
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
tail = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
binw = float(sys.argv[3]) if len(sys.argv) > 3 else 5.0
rows = db.execute("select name, start, end from kernels order by start").fetchall()
t_hi = max(r[2] for r in rows)
t_lo = t_hi - tail * 1e6
nb = int(tail / binw)
busy = [0.0] * nb
conc = [0.0] * nb
who = [defaultdict(float) for _ in range(nb)]
# union via sweep
ev = []
for n, s, e in rows:
    if e <= t_lo:
        continue
    s = max(s, t_lo)
    short = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-28:]
    b0, b1 = int((s - t_lo) / (binw * 1e6)), min(nb - 1, int((e - t_lo) / (binw * 1e6)))
    for b in range(b0, b1 + 1):
        lo, hi = t_lo + b * binw * 1e6, t_lo + (b + 1) * binw * 1e6
        ov = max(0.0, min(e, hi) - max(s, lo))
        conc[b] += ov
        who[b][short] += ov
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
act, cur = 0, t_lo
for t, d in ev:
    if act > 0 and t > cur:
        b0, b1 = int((cur - t_lo) / (binw * 1e6)), min(nb - 1, int((t - t_lo) / (binw * 1e6)))
        for b in range(b0, b1 + 1):
            lo, hi = t_lo + b * binw * 1e6, t_lo + (b + 1) * binw * 1e6
            busy[b] += max(0.0, min(t, hi) - max(cur, lo))
    cur = max(cur, t)
    act += d
for b in range(nb):
    top = max(who[b].items(), key=lambda kv: kv[1])[0] if who[b] else "-"
    print("%6.0f ms  busy %5.1f %%  kernels in flight %5.2f  %s" % (b * binw - tail, 100 * busy[b] / (binw * 1e6), conc[b] / (binw * 1e6), top))
