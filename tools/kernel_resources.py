#!/usr/bin/env python3
"""Registers, LDS and scratch of a kernel as rocprofv3's kernel trace records them.
usage: tools/kernel_resources.py <results.db> <kernel substring>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
want = [c for c in cols if any(k in c.lower() for k in ("vgpr", "sgpr", "lds", "scratch", "workgroup", "grid"))]
rows = db.execute("select name, %s, count(*), avg(duration) from kernels where name like ? group by name, grid_x" % ", ".join(want), ("%" + sys.argv[2] + "%",)).fetchall()
print(" | ".join(["kernel"] + want + ["launches", "avg_duration_ns"]))
for r in rows:
    print(" | ".join(str(x)[:60] for x in r))
