#!/usr/bin/env python3
"""SMPL forward (pose chain + blend GEMM + skinning, csrc/smpl.hip) alone at the benchmark batch: time per call.
    python tools/bench_smpl.py [B] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egohmr_amd.factory import build_synthetic_model  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
model = build_synthetic_model(dev, 0)
smpl = model.smpl
g = torch.Generator(device="cpu").manual_seed(0)
betas = torch.randn(B, 10, generator=g).to(dev)
R = torch.linalg.qr(torch.randn(B, 24, 3, 3, generator=g))[0].to(dev)
out = smpl(betas=betas, body_pose=R[:, 1:], global_orient=R[:, :1], pose2rot=False)
for _ in range(3):
    smpl(betas=betas, body_pose=R[:, 1:], global_orient=R[:, :1], pose2rot=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    smpl(betas=betas, body_pose=R[:, 1:], global_orient=R[:, :1], pose2rot=False)
e1.record()
torch.cuda.synchronize()
print(f"SMPL forward B={B}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per call (includes the pose chain and host glue); "
      f"|verts| checksum {float(out.vertices.abs().sum()):.6e}")
