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
        fb = m.backbone.folded(channels_last=False, matrix_core=True)
        t, o = timeit(lambda: fb(img))
        print(f"resnet50 split-f16 implicit GEMM : {t:7.2f} ms   max|d| vs shipped {float((o - ref).abs().max()):.2e}")
        for cl in (False, True):
            fb = m.backbone.folded(channels_last=cl, matrix_core=False)
            t, o = timeit(lambda: fb(img))
            print(f"resnet50 BN-folded cl={cl!s:5s}  : {t:7.2f} ms   max|d| vs shipped {float((o - ref).abs().max()):.2e}")

    # --- experiments: fewer elementwise passes around the convolutions
    bb = m.backbone

    def fold(conv, bn):
        scale = (bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps))
        return (conv.weight.double() * scale.view(-1, 1, 1, 1)).float(), (bn.bias.double() - bn.running_mean.double() * scale).float(), conv.stride, conv.padding

    stem = fold(bb.conv1, bb.bn1)
    blocks = []
    for layer in (bb.layer1, bb.layer2, bb.layer3, bb.layer4):
        for blk in layer:
            blocks.append((fold(blk.conv1, blk.bn1), fold(blk.conv2, blk.bn2), fold(blk.conv3, blk.bn3),
                           fold(blk.downsample[0], blk.downsample[1]) if blk.downsample is not None else None))

    def cr(x, p):      # MIOpen fusion plan: conv + bias + relu
        return torch.miopen_convolution_relu(x, p[0], p[1], p[2], p[3], (1, 1), 1)

    def car(x, p, z):  # conv + z + bias + relu
        return torch.miopen_convolution_add_relu(x, p[0], z, 1.0, p[1], p[2], p[3], (1, 1), 1)

    def run_fused(x):
        x = F.max_pool2d(cr(x, stem), 3, stride=2, padding=1)
        for c1, c2, c3, ds in blocks:
            y = cr(x, c1)
            y = cr(y, c2)
            sc = x if ds is None else F.conv2d(x, ds[0], ds[1], stride=ds[2], padding=ds[3])
            x = car(y, c3, sc)
        return x.mean(dim=(2, 3))

    try:
        t, o = timeit(lambda: run_fused(img))
        print(f"resnet50 miopen conv+bias+relu : {t:7.2f} ms   max|d| vs shipped {float((o - ref).abs().max()):.2e}")
    except Exception as e:  # noqa: BLE001
        print("miopen fused path failed:", repr(e)[:300])
