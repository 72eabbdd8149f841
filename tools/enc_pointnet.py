#!/usr/bin/env python3
"""Scene PointNet alone at the benchmark shape (csrc/linear.hip): time per forward and issued matrix-core rate.
    python tools/enc_pointnet.py [B] [N] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egohmr_amd.encoders import ResnetPointnet  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
net = ResnetPointnet().to(dev).eval()
pts = torch.randn(B, N, 3, device=dev)
for _ in range(2):
    net(pts)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    net(pts)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
H = 256
macs = B * N * (2 * H * H + (H + 32) * H + 3 * (H * H + 2 * H * H))
print(f"PointNet B={B} N={N}: {ms:.3f} ms per forward; {macs * 6 / ms / 1e9:.0f} TFLOP/s issued (3 MFMAs per product)")
