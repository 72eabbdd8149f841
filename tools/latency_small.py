#!/usr/bin/env python3
"""Latency of a small sampling call: B=8, DDIM-5 of 50, N=1024 scene points, conditioning cached - the regime where
the host's launch work matters.  Eager enqueue (5 launches per step) vs hipGraph replay of the captured loop.
    python tools/latency_small.py [B] [respacing]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egohmr_amd import synthetic as syn  # noqa: E402
from egohmr_amd.diffusion import create_gaussian_diffusion  # noqa: E402
from egohmr_amd.factory import batch_to_device, build_synthetic_model  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rs = sys.argv[2] if len(sys.argv) > 2 else "ddim5"
dev = torch.device("cuda:0")
model = build_synthetic_model(dev, 0)
d = create_gaussian_diffusion(num_diffusion_timesteps=50, timestep_respacing=rs)
b = batch_to_device(syn.make_batch(B, 1024, seed=1), dev)
noise = torch.from_numpy(syn.make_noise_stack(d.num_timesteps, B, seed=1)).to(dev)
fs = model.fused_sampler
res = {}
ref = None
for mode in (False, True):
    model.use_hip_graph = mode
    for _ in range(3):
        o = fs.run(d, b, noise, ddim=bool(rs))
    torch.cuda.synchronize()
    ts = []
    for _ in range(50):
        t0 = time.perf_counter()
        o = fs.run(d, b, noise, ddim=bool(rs))
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    res["hip_graph" if mode else "eager"] = {"p50_ms": ts[len(ts) // 2] * 1e3, "p10_ms": ts[len(ts) // 10] * 1e3, "min_ms": ts[0] * 1e3}
    v = o["other_outputs"]["pred_vertices"]
    if ref is None:
        ref = v.clone()
    else:
        res["graph_equals_eager"] = bool(torch.equal(ref, v))
print(json.dumps({"workload": f"B={B} {rs} of 50, N=1024, T={d.num_timesteps} steps, conditioning cached, f16x3 (all steps: T <= 10)", **res}))
