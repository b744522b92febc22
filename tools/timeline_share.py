#!/usr/bin/env python3
"""Who holds the GPU while many proofs are in flight: sweep over the kernel trace of a rocprofv3 run, at every moment split
the elapsed time equally among the kernels executing, and sum per kernel name.  Also the fraction of the window in which
no kernel ran.   usage: tools/timeline_share.py <results.db> [proofs_per_s] [tail_ms]  (the window is the last tail_ms of the trace)"""
import sqlite3, sys, heapq
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
rate = float(sys.argv[2]) if len(sys.argv) > 2 else 0
tail_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = db.execute("select name, start, end from kernels order by start").fetchall()
t_lo, t_hi = rows[0][1], max(r[2] for r in rows)
t_lo = t_hi - tail_ms * 1e6 if tail_ms else t_lo
ev = []
for name, s, e in rows:
    if e <= t_lo:
        continue
    n = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-60:]
    ev.append((max(s, t_lo), 1, n))
    ev.append((e, -1, n))
ev.sort()
active = defaultdict(int)
share = defaultdict(float)
excl = defaultdict(float)
idle = 0.0
cur = t_lo
nact = 0
for t, d, n in ev:
    dt = t - cur
    if dt > 0:
        if nact == 0:
            idle += dt
        else:
            for k, c in active.items():
                if c:
                    share[k] += dt * c / nact
        cur = t
    active[n] += d
    nact += d
wall = t_hi - t_lo
n_proofs = rate * wall / 1e9
print("window %.1f ms, idle (no kernel running) %.1f %%" % (wall / 1e6, 100 * idle / wall))
print("| kernel | share of window % | ms per proof |")
print("|---|---|---|")
for k, v in sorted(share.items(), key=lambda x: -x[1])[:30]:
    print("| %s | %.1f | %s |" % (k, 100 * v / wall, ("%.3f" % (v / 1e6 / n_proofs)) if n_proofs else "-"))
