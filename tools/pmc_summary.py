#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per kernel:  python tools/pmc_summary.py <dir> [kernel-substring]"""
import collections
import csv
import glob
import sys

d, key = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
agg = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{k:60s} {c:28s} n={len(v):4d} avg={sum(v) / len(v):16.1f}")
