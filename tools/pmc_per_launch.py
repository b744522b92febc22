#!/usr/bin/env python3
"""Per-launch averages of the PMC counters of one kernel, grouped by grid size where the view has one, from a rocprofv3 --pmc
rocpd database.      usage: tools/pmc_per_launch.py <results.db> <kernel substring>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2]
cols = [r[1] for r in db.execute("pragma table_info(pmc_events)").fetchall()]
grid = next((c for c in ("grid_size", "grid_x", "grid_size_x") if c in cols), None)
sel = "name, counter_name, %s, count(*), avg(counter_value)" % (grid or "0")
grp = "name, counter_name" + (", " + grid if grid else "")
rows = db.execute("select %s from pmc_events group by %s" % (sel, grp)).fetchall()
for name, ctr, g, cnt, avg in sorted(rows, key=lambda r: (r[1], r[2] or 0)):
    if pat in name:
        print("%-16s grid=%-9s %-22s launches=%-4d avg=%.1f" % (pat, g, ctr, cnt, avg))
