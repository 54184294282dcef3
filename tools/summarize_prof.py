#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel_stats / counter_collection) into a short text summary
with kernel names truncated, for committing under profiles/."""
import csv
import glob
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "")
    cut = name.find("(")
    if cut > 0:
        name = name[:cut]
    return name[:90]


def kernel_stats(d):
    for f in glob.glob(d + "/**/*_kernel_stats.csv", recursive=True):
        print(f"# kernel stats ({f.split('/')[-1]})")
        print(f"{'kernel':92s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
        for r in csv.DictReader(open(f)):
            print(f"{short(r['Name']):92s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.2f} "
                  f"{float(r['MinNs'])/1e3:10.2f} {float(r['MaxNs'])/1e3:10.2f} {float(r['Percentage']):6.2f}")


def counters(d):
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        acc = defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            key = (short(r["Kernel_Name"]), r["Counter_Name"])
            acc[key][0] += 1
            acc[key][1] += float(r["Counter_Value"])
        print(f"# counters ({f.split('/')[-1]}): per-dispatch mean")
        for (k, c), (cnt, tot) in sorted(acc.items()):
            print(f"{k:92s} {c:14s} dispatches={cnt:5d} mean={tot/cnt:16.3f}")


if __name__ == "__main__":
    for d in sys.argv[1:]:
        kernel_stats(d)
        counters(d)
