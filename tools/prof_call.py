#!/usr/bin/env python3
"""N whole sampling calls (encoders + DDPM-100 loop, B=256) back to back, for rocprofv3 --kernel-trace: tools/gaps.py then lists the
idle gaps of the GPU timeline."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egohmr_amd import synthetic as syn  # noqa: E402
from egohmr_amd.diffusion import create_gaussian_diffusion  # noqa: E402
from egohmr_amd.factory import batch_to_device, build_synthetic_model  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
model = build_synthetic_model(dev, 0)
d = create_gaussian_diffusion(num_diffusion_timesteps=100, timestep_respacing="")
b = batch_to_device(syn.make_batch(256, 4096, seed=100), dev)
noise = torch.from_numpy(syn.make_noise_stack(100, 256, seed=100)).to(dev)
fs = model.fused_sampler
for _ in range(n):
    fs.invalidate()
    fs.run(d, b, noise, ddim=False)
torch.cuda.synchronize()
