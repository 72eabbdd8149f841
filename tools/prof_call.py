#!/usr/bin/env python3
"""N whole sampling calls (encoders + loop, B = 256, trained-like weights, every step split-f16) back to back, for rocprofv3 --kernel-trace:
    rocprofv3 --kernel-trace --output-format csv -d <dir> -o kt -- python tools/prof_call.py [n_calls] [ddpm100|c2_ddim10]
tools/gaps.py then lists the idle gaps of the GPU timeline, tools/call_timeline.py the kernels of the last call in order."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egohmr_amd import synthetic as syn  # noqa: E402
from egohmr_amd.diffusion import create_gaussian_diffusion  # noqa: E402
from egohmr_amd.factory import batch_to_device, build_synthetic_model  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
wl = sys.argv[2] if len(sys.argv) > 2 else "ddpm100"
rs = {"ddpm100": "", "c2_ddim10": "ddim10"}[wl]
dev = torch.device("cuda:0")
model = build_synthetic_model(dev, 0, sensitive=dict(num_diffusion_timesteps=100))
model.f16x3_last_steps = None                      # every step split-f16 (what the calibration picks for these weights): no calibration launches in the trace
d = create_gaussian_diffusion(num_diffusion_timesteps=100, timestep_respacing=rs)
T = d.num_timesteps
b = batch_to_device(syn.make_batch(256, 4096, seed=100), dev)
noise = torch.from_numpy(syn.make_noise_stack(T, 256, seed=100)).to(dev)
fs = model.fused_sampler
for _ in range(n):
    fs.invalidate()
    fs.run(d, b, noise, ddim=bool(rs), defer_status=True)
torch.cuda.synchronize()
fs.check_status()
