#!/usr/bin/env python3
"""Per-kernel average of one PMC counter from a rocprofv3 --pmc rocpd database.
usage: tools/pmc_stats.py <results.db>   -> lines "kernel launches avg_counter_value"."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, counter_name, count(*), avg(counter_value) from pmc_events group by name, counter_name "
                      "order by sum(counter_value) desc").fetchall()
    for name, ctr, cnt, avg in rows[:28]:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-40:]
        print("%-40s %-12s launches=%-5d avg=%.1f" % (short, ctr, cnt, avg))


if __name__ == "__main__":
    main()
