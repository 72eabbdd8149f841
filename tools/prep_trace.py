#!/usr/bin/env python3
"""The kernels of ONE FusedSampler.prepare + run (DDIM-10) in launch order with their durations, from a rocprofv3 --kernel-trace CSV written by
running this script under rocprofv3:   rocprofv3 --kernel-trace --output-format csv -d OUT -o kt -- python tools/prep_trace.py run
                                       python tools/prep_trace.py show OUT [min_us]      (everything except the big conv / GEMM / chain kernels)"""
import csv
import glob
import os
import sys

if sys.argv[1] == "run":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from egohmr_amd import synthetic as syn
    from egohmr_amd.diffusion import create_gaussian_diffusion
    from egohmr_amd.factory import batch_to_device, build_synthetic_model
    dev = torch.device("cuda:0")
    model = build_synthetic_model(dev, 0, sensitive=dict(num_diffusion_timesteps=100))
    model.f16x3_last_steps = None
    d = create_gaussian_diffusion(num_diffusion_timesteps=100, timestep_respacing="ddim10")
    b = batch_to_device(syn.make_batch(256, 4096, seed=100), dev)
    noise = torch.from_numpy(syn.make_noise_stack(10, 256, seed=100)).to(dev)
    fs = model.fused_sampler
    for i in range(4):
        fs.invalidate()
        torch.cuda.synchronize()
        fs.run(d, b, noise, ddim=True, defer_status=True)
    torch.cuda.synchronize()
else:
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
    last = max(i for i, r in enumerate(rows) if "item_prep_kernel" in r[2])        # the last call starts a few kernels before its item_prep_kernel
    seg = rows[max(0, last - 3):]
    t0 = seg[0][0]
    big = ("conv_x2_tile", "linear_tile", "gcn_hidden_chain", "stem_mfma")
    tot_small = 0.0
    for s, e, n in seg:
        us = (e - s) / 1e3
        if any(k in n for k in big):
            continue
        tot_small += us
        if us >= min_us:
            print(f"{(s - t0) / 1e3:9.1f} us  {us:7.1f} us  {n[:110]}")
    print(f"call span {(seg[-1][1] - t0) / 1e6:.3f} ms; kernels other than convs / GEMMs / chain: {tot_small / 1e3:.3f} ms")
