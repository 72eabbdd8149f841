#!/usr/bin/env python3
"""Debug aid for the one-launch loop: run it for several step counts / batch sizes under EHM_LOOP_DEBUG=1 (prints launch time + counters)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egohmr_amd import synthetic as syn
from egohmr_amd.diffusion import create_gaussian_diffusion
from egohmr_amd.factory import batch_to_device, build_synthetic_model
dev = torch.device("cuda:0")
model = build_synthetic_model(dev, 0, diffuse_fuse=True, sensitive=dict(num_diffusion_timesteps=100))
model.f16x3_last_steps = None
for B, rs in [(256, "ddim5"), (256, "ddim10"), (256, "ddim10"), (128, "ddim10"), (256, "ddim20"), (256, "")]:
    d = create_gaussian_diffusion(num_diffusion_timesteps=100, timestep_respacing=rs)
    T = d.num_timesteps
    batch = batch_to_device(syn.make_batch(B, 512, seed=100), dev)
    noise = torch.from_numpy(syn.make_noise_stack(T, B, seed=100)).to(dev)
    print("==== B", B, "T", T, file=sys.stderr, flush=True)
    try:
        model.fused_sampler.invalidate()
        model.fused_sampler.run(d, batch, noise, ddim=bool(rs))
        torch.cuda.synchronize()
    except Exception as e:
        print("ERR", str(e)[:80], file=sys.stderr, flush=True)
