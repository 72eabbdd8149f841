#!/usr/bin/env python3
"""Effective shader clock of a kernel = GRBM_GUI_ACTIVE (cycles the GPU was busy) / kernel duration, from a rocprofv3 pass that collected
the counter together with --kernel-trace:  python tools/pmc_clock.py <dir of the pass> [kernel-substring]"""
import csv
import glob
import sys

d, key = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "gcn_hidden_chain")
cyc = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cyc.setdefault(r["Dispatch_Id"], float(r["Counter_Value"]))
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
pairs = [(cyc[k], dur[k]) for k in cyc if k in dur]
if pairs:
    c = sum(p[0] for p in pairs) / len(pairs)
    t = sum(p[1] for p in pairs) / len(pairs)
    # the counter is summed over the 8 XCDs of an MI355X
    print(f"{key}: GRBM_GUI_ACTIVE {c:.0f} cycles (sum over 8 XCDs) over {t * 1e6:.1f} us -> {c / 8 / t / 1e9:.3f} GHz effective shader clock ({len(pairs)} dispatches)")
else:
    print("no matching dispatches (counter and kernel trace in the same pass?)")
