#!/usr/bin/env python3
"""HBM-side bytes per hidden conv from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh:
    python tools/pmc_traffic.py gpurun_out/<tag>  > profiles/pmc_traffic.json
FETCH_SIZE is reported in KiB... on gfx950 the counter undercounts reads by 2x (MI355X_MICROARCH.md, HBM / rocprofv3 section:
calibrated in round 1 on gcn_out_dot_kernel, 24.8 MB raw vs 48.0 MB algorithmic), WRITE_SIZE is exact; both in units of 1 KiB.
The chained launch runs 8 convs: per-conv = per-launch / 8."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

d = sys.argv[1]
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sha = hashlib.sha1(open(os.path.join(repo, "egohmr_amd", "csrc", "gcn_tile.hip"), "rb").read()).hexdigest()
out = {}
for prec, tmpl in (("f16", "gcn_hidden_chain_kernel<1, 8>"), ("f16x3", "gcn_hidden_chain_kernel<3, 4>")):
    vals = collections.defaultdict(list)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(f"{d}/pmc_{prec}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if tmpl in r["Kernel_Name"] and r["Counter_Name"] == c:
                    vals[c].append(float(r["Counter_Value"]))
    if vals["FETCH_SIZE"] and vals["WRITE_SIZE"]:
        fetch = sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"]) * 1024 * 2      # gfx950 correction
        write = sum(vals["WRITE_SIZE"]) / len(vals["WRITE_SIZE"]) * 1024
        out[prec] = {"kernel": tmpl, "kernel_source_sha1": sha, "fetch_bytes_per_launch_corrected_x2": fetch, "write_bytes_per_launch": write,
                     "bytes_per_launch": fetch + write, "bytes_per_conv": (fetch + write) / 8, "launches_averaged": len(vals["FETCH_SIZE"]),
                     "workload": "tools/bench_hidden.py (EHM_STACK=1): B=256 x 2 passes, relu-like activations (half zeros), as in the sampler"}
print(json.dumps(out, indent=1))
