#!/usr/bin/env python3
"""Where the waves of each kernel spend their cycles: one rocprofv3 --pmc pass with SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS (MI355X_MICROARCH.md: WAIT_ANY = parked on s_waitcnt / a barrier,
WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY = issuing; the three are disjoint and add up to about WAVE_CYCLES).
usage: tools/pmc_wave_cycles.py <results.db> [kernels to skip, substring, comma separated]"""
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
skip = [s for s in (sys.argv[2].split(",") if len(sys.argv) > 2 else []) if s]
rows = db.execute("select name, counter_name, count(*), sum(counter_value) from pmc_events group by name, counter_name").fetchall()
d = defaultdict(dict)
for name, c, k, v in rows:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-44:]
    if any(s in short for s in skip):
        continue
    d[short][c] = v
    d[short]["launches"] = k
tot = sum(v.get("SQ_WAVE_CYCLES", 0) for v in d.values()) or 1
print("| kernel | launches | wave cycles, % of the run | issuing (ACTIVE_INST_ANY) | of which VALU | parked: s_waitcnt / barrier (WAIT_ANY) | issue stall (WAIT_INST_ANY) | of which LDS | VALU instructions per wave-cycle issuing VALU |")
print("|---|---|---|---|---|---|---|---|---|")
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:16]:
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    f = lambda c: 100.0 * v.get(c, 0) / wc
    av = v.get("SQ_ACTIVE_INST_VALU", 0) or 1
    print("| %s | %d | %.1f | %.1f %% | %.1f %% | %.1f %% | %.1f %% | %.1f %% | %.3f |" % (k, v["launches"], 100.0 * wc / tot, f("SQ_ACTIVE_INST_ANY"), f("SQ_ACTIVE_INST_VALU"), f("SQ_WAIT_ANY"),
                                                                                  f("SQ_WAIT_INST_ANY"), f("SQ_WAIT_INST_LDS"), v.get("SQ_INSTS_VALU", 0) / av))
