#!/usr/bin/env python3
"""Every dispatch of the LAST sampling call of a rocprofv3 --kernel-trace CSV (tools/prof_call.py) up to the first chained conv launch, unmerged:
    python tools/call_dispatches.py <dir>          start (us), duration (us), grid, kernel"""
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size", "")) for r in csv.DictReader(open(f))]
rows.sort()
starts = [i for i, r in enumerate(rows) if "item_prep_kernel" in r[2]]
rows = rows[starts[-1]:]
t0 = rows[0][0]
for s, e, n, g in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n).split("(")[0][:60]
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f} {g:>8} {n}")
    if "gcn_hidden_chain" in n:
        break
