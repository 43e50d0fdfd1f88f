#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output dirs (kernel stats + per-kernel PMC means) into a small text report."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return name[:90]


def main(root):
    for path in sorted(glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True)):
        print("== kernel stats:", os.path.relpath(path, root))
        with open(path) as f:
            rows = list(csv.DictReader(f))
        for r in rows[:14]:
            print("  {:<92s} calls {:>5s} total_ns {:>13s} avg_ns {:>12s} pct {:>6s}".format(
                short(r.get("Name", "")), r.get("Calls", ""), r.get("TotalDurationNs", ""), r.get("AverageNs", ""), r.get("Percentage", "")))
    for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        print("== counters:", os.path.relpath(path, root))
        acc = defaultdict(lambda: defaultdict(list))
        with open(path) as f:
            for r in csv.DictReader(f):
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in sorted(acc):
            vals = "  ".join("{}={:.4g}".format(c, sum(v) / len(v)) for c, v in sorted(acc[k].items()))
            print("  {:<70s} n={:<4d} {}".format(k[:70], len(next(iter(acc[k].values()))), vals))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out")
