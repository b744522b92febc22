#!/usr/bin/env python3
"""Per-kernel totals of the PMC counters of a rocprofv3 --pmc run, divided by the number of proofs.
usage: tools/pmc_per_proof.py <results.db> <n_proofs>"""
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
n = float(sys.argv[2])
rows = db.execute("select name, counter_name, count(*), sum(counter_value) from pmc_events group by name, counter_name").fetchall()
d = defaultdict(dict)
ctrs = []
for name, c, k, v in rows:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-44:]
    d[short][c] = v / n
    if c not in ctrs:
        ctrs.append(c)
tot = {c: sum(v.get(c, 0) for v in d.values()) or 1 for c in ctrs}
print("| kernel | " + " | ".join("%s per proof | %%" % c for c in ctrs) + " |")
print("|---|" + "---|---|" * len(ctrs))
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get(ctrs[0], 0))[:32]:
    print("| %s | " % k + " | ".join("%.3e | %.1f" % (v.get(c, 0), 100 * v.get(c, 0) / tot[c]) for c in ctrs) + " |")
print("| total | " + " | ".join("%.3e | 100" % tot[c] for c in ctrs) + " |")
