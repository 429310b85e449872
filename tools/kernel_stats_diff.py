#!/usr/bin/env python3
"""Two rocprofv3 `--kernel-trace --stats` kernel_stats.csv files side by side: per kernel the average duration in A and in B and
their ratio, sorted by the total time gained (what a different input or build costs, kernel by kernel).
    python tools/kernel_stats_diff.py a_kernel_stats.csv b_kernel_stats.csv [min total us]"""
import csv
import re
import sys


def load(path):
    out = {}
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "").strip()
        calls, total = int(r["Calls"]), float(r["TotalDurationNs"])
        c, t = out.get(name, (0, 0.0))
        out[name] = (c + calls, t + total)
    return out


def main():
    a, b = load(sys.argv[1]), load(sys.argv[2])
    floor = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 0.0
    rows = []
    for name in sorted(set(a) | set(b)):
        ca, ta = a.get(name, (0, 0.0))
        cb, tb = b.get(name, (0, 0.0))
        if max(ta, tb) < floor:
            continue
        rows.append((tb - ta, name, ca, ta, cb, tb))
    print(f"{'kernel':58s} {'calls':>6s} {'A avg us':>10s} {'B avg us':>10s} {'B/A':>6s} {'B-A total ms':>13s}")
    for d, name, ca, ta, cb, tb in sorted(rows, reverse=True):
        aa, ab = (ta / ca / 1e3 if ca else 0.0), (tb / cb / 1e3 if cb else 0.0)
        print(f"{name[:58]:58s} {max(ca, cb):6d} {aa:10.1f} {ab:10.1f} {(ab / aa if aa else 0):6.2f} {d / 1e6:13.3f}")
    print(f"{'total':58s} {'':6s} {'':10s} {'':10s} {'':6s} {sum(r[0] for r in rows) / 1e6:13.3f}")


if __name__ == "__main__":
    main()
