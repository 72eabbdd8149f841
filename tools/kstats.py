#!/usr/bin/env python3
"""Print the top rows of a rocprofv3 *_kernel_stats.csv:  python tools/kstats.py <csv or dir> [n]"""
import csv
import glob
import os
import sys

path = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
if os.path.isdir(path):
    path = glob.glob(path + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"{path}: total GPU time {tot / 1e6:.2f} ms over {sum(int(r['Calls']) for r in rows)} dispatches")
for r in rows[:n]:
    print(f"{r['Name'][:86]:86s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs']) / 1e3:9.1f} tot_ms={int(r['TotalDurationNs']) / 1e6:8.2f} {r['Percentage']:>6s}%")
