#!/usr/bin/env python3
"""Time the two conditioning encoders at the benchmark shape (B=256): python tools/bench_encoders.py"""
import os
import sys
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egohmr_amd.factory import build_synthetic_model  # noqa: E402

dev = torch.device("cuda:0")
m = build_synthetic_model(dev, 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
img = torch.randn(B, 3, 224, 224, device=dev)
pts = torch.rand(B, 4096, 3, device=dev) * 2 - 1


def timeit(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3, out


with torch.no_grad():
    t, ref = timeit(lambda: m.backbone(img))
    print(f"resnet50 as shipped            : {t:7.2f} ms")
    t, _ = timeit(lambda: m.scene_enc(pts))
    print(f"pointnet (HIP split-f16)      : {t:7.2f} ms")
    if hasattr(m.backbone, "fold_batchnorm"):
        for cl in (False, True):
            fb = m.backbone.folded(channels_last=cl)
            t, o = timeit(lambda: fb(img))
            print(f"resnet50 BN-folded cl={cl!s:5s}  : {t:7.2f} ms   max|d| vs shipped {float((o - ref).abs().max()):.2e}")
