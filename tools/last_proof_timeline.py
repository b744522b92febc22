#!/usr/bin/env python3
"""Kernel timeline of the last proof in a rocprofv3 rocpd database (single proof in flight): offset from the first kernel,
duration and the idle gap before each launch; consecutive launches of one kernel are merged.
usage: tools/last_proof_timeline.py <results.db> [min_us]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
rows = db.execute("select name,start,duration,grid_x,workgroup_x from kernels order by start").fetchall()
last = [i for i, r in enumerate(rows) if 'k_sh_w' in r[0]]
beg, end = last[-2] + 1, last[-1] + 1
t0 = rows[beg][1]
prev_end = t0
busy = 0.0
out = []
for r in rows[beg:end]:
    n = r[0].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    gap = (r[1] - prev_end) / 1e3
    busy += r[2] / 1e3
    out.append([n, (r[1] - t0) / 1e3, r[2] / 1e3, gap, r[3] // max(1, r[4]), 1])
    prev_end = max(prev_end, r[1] + r[2])
merged = []
for o in out:
    if merged and merged[-1][0] == o[0] and o[3] < 20:
        merged[-1][2] += o[2]
        merged[-1][5] += 1
    else:
        merged.append(o)
print("%-36s %10s %9s %9s %8s" % ("kernel", "at_us", "dur_us", "gap_us", "blocks"))
for n, at, dur, gap, blocks, cnt in merged:
    if dur >= min_us or gap >= 50:
        print("%-36s %10.0f %9.1f %9.1f %8d%s" % (n[-36:], at, dur, gap, blocks, " x%d" % cnt if cnt > 1 else ""))
print("span %.1f us, kernel busy %.1f us, idle %.1f us" % ((prev_end - t0) / 1e3, busy, (prev_end - t0) / 1e3 - busy))
