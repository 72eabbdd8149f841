#!/usr/bin/env python3
"""Derived figures of tools/pmc_insitu.sh's passes for one kernel: duration (from the kernel trace of every pass), effective clock, MFMA-pipe busy
fraction, wait fraction, LDS instruction share.   python tools/pmc_insitu_derive.py <dir> [kernel-substring]"""
import csv
import glob
import sys

d, key = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "gcn_hidden_chain")
val, dur = {}, {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"]:
            val.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    p = f[len(d):].strip("/").split("/")[0]
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"]:
            dur.setdefault(p, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
avg = {k: sum(v) / len(v) for k, v in val.items()}
for p, v in sorted(dur.items()):
    print(f"{key}: duration under the {p} pass: {sum(v) / len(v):8.1f} us avg over {len(v)} launches (min {min(v):.1f}, max {max(v):.1f})")
g = avg.get("GRBM_GUI_ACTIVE")
t = dur.get("GRBM_GUI_ACTIVE")
if g and t:
    tt = sum(t) / len(t) * 1e-6
    clk = g / 8 / tt
    print(f"effective shader clock {clk / 1e9:.3f} GHz (GRBM_GUI_ACTIVE summed over 8 XCDs / duration)")
    # SQ counters are summed over the SEs' SQs; per-CU-cycle normalisation: cycles per XCD x 256 CUs x 4 SIMDs
    cu_cycles = g / 8 * 256
    if "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
        print(f"MFMA pipe busy: SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 256 CUs x 4 SIMDs) = {avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (cu_cycles * 4):.3f}"
              f"  (busy x clock = {avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (cu_cycles * 4) * clk / 1e9:.3f} GHz)")
if "SQ_WAVE_CYCLES" in avg:
    for c in ("SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_LDS"):
        if c in avg:
            print(f"{c} / SQ_WAVE_CYCLES = {avg[c] / avg['SQ_WAVE_CYCLES']:.3f}")
if "SQ_BUSY_CYCLES" in avg and "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
    print(f"SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES = {avg['SQ_VALU_MFMA_BUSY_CYCLES'] / avg['SQ_BUSY_CYCLES']:.3f} (raw ratio; units differ per counter)")
if "SQ_INSTS_LDS" in avg and "SQ_INSTS_VALU_MFMA_MOPS_F16" in avg:
    print(f"LDS instructions per launch {avg['SQ_INSTS_LDS']:.3e}; MFMA MOPS_F16 {avg['SQ_INSTS_VALU_MFMA_MOPS_F16']:.3e}")
