#!/usr/bin/env python3
"""Smoke over extreme batch shapes (B = 1 ... 2048, ragged N): shapes, finiteness, time, peak memory.  python tools/check_sizes.py"""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egohmr_amd import synthetic as syn
from egohmr_amd.diffusion import create_gaussian_diffusion
from egohmr_amd.factory import batch_to_device, build_synthetic_model
dev = torch.device("cuda:0")
model = build_synthetic_model(dev, 0)
d = create_gaussian_diffusion(num_diffusion_timesteps=50, timestep_respacing="ddim5")
for B, N in [(1, 100), (5, 4097), (1024, 4096), (2048, 1024)]:
    b = batch_to_device(syn.make_batch(B, N, seed=1), dev)
    noise = torch.from_numpy(syn.make_noise_stack(d.num_timesteps, B, seed=1)).to(dev)
    torch.cuda.synchronize(); t = time.perf_counter()
    out = d.val_losses(model, b, shape=[B, 144], clip_denoised=False, timestep_respacing="ddim5", compute_loss=False, noise_stack=noise)
    torch.cuda.synchronize()
    v = out["pred_vertices"]
    print(B, N, tuple(v.shape), bool(torch.isfinite(v).all()), f"{(time.perf_counter()-t)*1e3:.1f} ms", f"peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
# consistency: item 0 of B=5 equals B=1 run with the same inputs?
