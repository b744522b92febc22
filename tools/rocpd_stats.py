#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace) as a per-kernel stats table (markdown/CSV-ish).
usage: tools/rocpd_stats.py <results.db> [> profiles/xxx.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(lds_size), max(scratch_size) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total_ms | avg_us | min_us | max_us | % | vgpr | lds | scratch |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        name = r[0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-70:]
        print("| %s | %d | %.3f | %.1f | %.1f | %.1f | %.1f | %s | %s | %s |" % (name, r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot, r[6], r[7], r[8]))


if __name__ == "__main__":
    main()
