#!/usr/bin/env python3
"""The kernels of the LAST sampling call of a rocprofv3 --kernel-trace CSV (tools/prof_call.py), in start order, consecutive launches of one
kernel merged:  python tools/call_timeline.py <dir> [n_calls]   (the trace is cut into n_calls equal runs by the stem kernel's launches)"""
import csv
import glob
import re
import sys

d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
starts = [i for i, r in enumerate(rows) if "item_prep_kernel" in r[2]]
rows = rows[starts[-1]:]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:70]


t0 = rows[0][0]
out, prev_end = [], rows[0][0]
for s, e, n in rows:
    n = short(n)
    gap = max(0, s - prev_end)
    if out and out[-1][0] == n:
        out[-1][1] += 1; out[-1][2] += e - s; out[-1][3] += gap; out[-1][5] = e
    else:
        out.append([n, 1, e - s, gap, s, e])
    prev_end = max(prev_end, e)
tot_k = sum(o[2] for o in out); tot_gap = sum(o[3] for o in out)
print(f"last call: {(prev_end - t0) / 1e6:.3f} ms wall, kernels {tot_k / 1e6:.3f} ms, gaps {tot_gap / 1e6:.3f} ms, {sum(o[1] for o in out)} dispatches")
for n, c, dur, gap, s, e in out:
    print(f"  t={(s - t0) / 1e3:9.1f} us  {n:70s} x{c:<4d} {dur / 1e3:9.1f} us  gaps {gap / 1e3:7.1f} us")
agg = {}
for n, c, dur, gap, s, e in out:
    a = agg.setdefault(n, [0, 0, 0]); a[0] += c; a[1] += dur; a[2] += gap
print("by kernel:")
for n, (c, dur, gap) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:70s} x{c:<4d} {dur / 1e3:9.1f} us  (+ gaps in front {gap / 1e3:7.1f} us)")
