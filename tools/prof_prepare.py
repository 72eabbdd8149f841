#!/usr/bin/env python3
"""FusedSampler.prepare alone (encoders + projections), for rocprofv3 --kernel-trace --stats: 2 warm-up + N timed calls."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egohmr_amd import synthetic as syn  # noqa: E402
from egohmr_amd.factory import batch_to_device, build_synthetic_model  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
model = build_synthetic_model(dev, 0)
b = batch_to_device(syn.make_batch(256, 4096, seed=100), dev)
fs = model.fused_sampler
for _ in range(n):
    fs.invalidate()
    fs.prepare(b)
torch.cuda.synchronize()
