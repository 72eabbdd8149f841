#!/usr/bin/env python3
"""Idle gaps of the GPU timeline from a rocprofv3 --kernel-trace CSV:  python tools/gaps.py <dir> [min_gap_us] [last_ms]
Looks at the last `last_ms` milliseconds of the trace (steady state), prints busy / idle totals and the largest gaps with their neighbours."""
import csv
import glob
import sys

d = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
last_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 85.0
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
cut = t1 - last_ms * 1e6
rows = [r for r in rows if r[0] >= cut]
busy, gaps, cur_end, prev = 0, [], rows[0][0], rows[0][2]
for s, e, name in rows:
    if s > cur_end:
        gaps.append(((s - cur_end) / 1e3, prev, name))
        busy += e - s
        cur_end = e
    else:
        if e > cur_end:
            busy += e - cur_end
            cur_end = e
    prev = name
span = (cur_end - rows[0][0]) / 1e3
idle = sum(g[0] for g in gaps)
print(f"window {span / 1e3:.2f} ms: busy {busy / 1e6:.2f} ms, idle {idle / 1e3:.2f} ms in {len(gaps)} gaps; gaps >= {min_gap} us:")
for g, a, b in sorted(gaps, reverse=True)[:25]:
    if g >= min_gap:
        print(f"  {g:8.1f} us  after {a}  before {b}")
small = [g[0] for g in gaps if g[0] < min_gap]
print(f"  ({len(small)} gaps below {min_gap} us: total {sum(small) / 1e3:.2f} ms, mean {sum(small) / max(1, len(small)):.2f} us)")
