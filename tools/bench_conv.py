#!/usr/bin/env python3
"""Per-shape timing of ehm_conv_nhwc_split over the ResNet-50 bottleneck convolutions at B=256:  python tools/bench_conv.py [B]
Prints algorithmic TFLOP/s and the HBM floor (input + output + identity bytes at 4 TB/s) next to the measured time."""
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egohmr_amd import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
L = _lib.lib()
# (H, Ci, Co, k, stride, has_residual, count) of torchvision's ResNet-50 v1.5 behind the stem
shapes = [(56, 64, 64, 1, 1, 0, 1), (56, 64, 64, 3, 1, 0, 3), (56, 64, 256, 1, 1, 1, 3), (56, 64, 256, 1, 1, 0, 1), (56, 256, 64, 1, 1, 0, 2),
          (56, 256, 128, 1, 1, 0, 1), (56, 128, 128, 3, 2, 0, 1), (28, 128, 512, 1, 1, 1, 4), (56, 256, 512, 1, 2, 0, 1), (28, 512, 128, 1, 1, 0, 3), (28, 128, 128, 3, 1, 0, 3),
          (28, 512, 256, 1, 1, 0, 1), (28, 256, 256, 3, 2, 0, 1), (14, 256, 1024, 1, 1, 1, 6), (28, 512, 1024, 1, 2, 0, 1), (14, 1024, 256, 1, 1, 0, 5), (14, 256, 256, 3, 1, 0, 5),
          (14, 1024, 512, 1, 1, 0, 1), (14, 512, 512, 3, 2, 0, 1), (7, 512, 2048, 1, 1, 1, 3), (14, 1024, 2048, 1, 2, 0, 1), (7, 2048, 512, 1, 1, 0, 2), (7, 512, 512, 3, 1, 0, 2)]
tot = tot_floor = 0.0
for H, Ci, Co, k, s, has_res, cnt in shapes:
    pad = k // 2
    Ho = (H + 2 * pad - k) // s + 1
    x = torch.relu(torch.randn(B, H, H, Ci, device=dev))
    K = k * k * Ci
    Cop = (Co + 127) // 128 * 128
    w2 = torch.zeros(Cop, K, device=dev)
    w2[:Co] = torch.randn(Co, K, device=dev) * (1.0 / math.sqrt(K))
    scale = 2.0 ** math.floor(math.log2(2048.0 / float(w2.abs().max())))
    buf = torch.empty(Cop, K, device=dev)
    _lib.check(L.ehm_split_pack(w2.data_ptr(), buf.data_ptr(), Cop, K, K, scale, None))
    bias = torch.randn(Co, device=dev)
    y = torch.empty(B, Ho, Ho, Co, device=dev)
    res = torch.randn_like(y) if has_res else None
    d = _lib.ConvDesc(x.data_ptr(), buf.data_ptr(), bias.data_ptr(), res.data_ptr() if has_res else None, y.data_ptr(), B, H, H, Ci, Co, k, k, s, pad, 1, scale)
    for _ in range(2):
        _lib.check(L.ehm_conv_nhwc_split(C.byref(d), None))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        _lib.check(L.ehm_conv_nhwc_split(C.byref(d), None))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 2.0 * B * Ho * Ho * K * Co
    by = 4.0 * (x.numel() + y.numel() * (2 if has_res else 1))
    floor = by / 4e12 * 1e3
    tot += ms * cnt
    tot_floor += max(floor, fl / 700e12 * 1e3) * cnt
    print(f"H{H:3d} {Ci:4d}->{Co:4d} k{k} s{s} res{has_res} x{cnt}: {ms * 1e3:7.1f} us  {fl / ms / 1e9:6.1f} TFLOP/s  {by / ms / 1e9:6.2f} TB/s(alg)  floor {floor * 1e3:6.1f} us")
print(f"sum over the network: {tot:.2f} ms   (max(HBM floor, 700 TFLOP/s) sum: {tot_floor:.2f} ms)")
