#!/usr/bin/env python3
"""Reads the traced runs of the driver's command in a directory (bench.py with ZKFHE_TRACE=1: *.json + *.err) and prints, per run, the rate
and where the timed wave's proofs were at which time: the last proof's phase-0 commitment, first challenge and phase-1 enqueue (ms after
the start of the timed region) -- a straggler there ends the wave late."""
import collections, glob, json, os, re, sys


def load(f):
    reg, rows = None, []
    for l in open(f, errors="replace"):
        m = re.match(r"\[zkfhe trace (\d+)\]\s+([\d.]+) ms \(\+\s*([\d.]+), cpu\s+([\d.]+)\) (.*) @([\d.]+)", l)
        if m:
            rows.append((float(m.group(6)), m.group(1), float(m.group(3)), float(m.group(4)), m.group(5)))
        m = re.match(r"\[bench trace\] timed region @([\d.]+) \.\. @([\d.]+)", l)
        if m:
            reg = (float(m.group(1)), float(m.group(2)))
    return reg, rows


def main(d):
    by_variant = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(d, "*.json"))):
        try:
            j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        except Exception as e:
            print(f, "ERR", e)
            continue
        reg, rows = load(f[:-5] + ".err")
        name = os.path.basename(f)[:-5]
        line = "%-22s %7.2f proofs/s" % (name, j["value"])
        if reg:
            T0, T1 = reg
            ph = collections.defaultdict(list)
            for t, tid, wall, cpu, what in rows:
                if T0 - 1 <= t <= T1 + 1:
                    ph[what].append((t - T0, wall, cpu))
            def last(k):
                return max((x[0] for x in ph.get(k, [(0, 0, 0)])), default=0)
            def mean_cpu(k):
                v = ph.get(k, [])
                return sum(x[2] for x in v) / max(1, len(v))
            line += "  region %5.1f ms; last: phase-0 commitment back %5.1f, rlc context %5.1f, phase-1 enqueue %5.1f, grand products %5.1f, quotient %5.1f; cpu of 'blind + upload phase 0' %4.2f ms" % (
                T1 - T0, last("commit phase 0 (GPU)"), last("rlc context"), last("gpu phase 1 (enqueue)"), last("grand products + commits"), last("quotient"), mean_cpu("blind + upload phase 0"))
        print(line)
        by_variant[re.sub(r"_\d+$", "", name)].append(j["value"])
    for k, v in by_variant.items():
        print("%-20s n=%d  mean %.1f  min %.1f  max %.1f" % (k, len(v), sum(v) / len(v), min(v), max(v)))


if __name__ == "__main__":
    main(sys.argv[1])
