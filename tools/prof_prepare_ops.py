#!/usr/bin/env python3
"""torch.profiler view of FusedSampler.prepare: which ATen ops / copies surround the HIP kernels (launch-bound host glue)."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egohmr_amd import synthetic as syn  # noqa: E402
from egohmr_amd.factory import batch_to_device, build_synthetic_model  # noqa: E402

dev = torch.device("cuda:0")
model = build_synthetic_model(dev, 0)
b = batch_to_device(syn.make_batch(256, 4096, seed=100), dev)
fs = model.fused_sampler
for _ in range(3):
    fs.invalidate()
    fs.prepare(b)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    fs.invalidate()
    fs.prepare(b)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="count", row_limit=30, max_name_column_width=60))
