#!/usr/bin/env python3
"""Per-convolution timing of the ResNet-50 trunk at the benchmark batch (csrc/conv.hip), with the two floors of each conv:
algorithmic HBM bytes (float32 activations in + out (+ identity), each once) and split-f16 matrix-core flops (3 MFMAs per product).
    python tools/enc_layers.py [B] [reps]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egohmr_amd import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
X2 = (sys.argv[3] if len(sys.argv) > 3 else "x2") == "x2"      # x2: activations in the split format (conv_x2_tile_kernel); f32: conv_nhwc_split_kernel
dev = torch.device("cuda:0")
L = _lib.lib()

# (name, H_in, Ci, Co, k, stride, has_res, count)
convs = []
H, cin = 56, 64
for li, (width, n, stride) in enumerate([(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)], 1):
    for b in range(n):
        s = stride if b == 0 else 1
        convs.append((f"l{li}b{b}.c1", H, cin, width, 1, 1, False))
        convs.append((f"l{li}b{b}.c2", H, width, width, 3, s, False))
        Ho = H // s
        if b == 0:
            convs.append((f"l{li}b{b}.ds", H, cin, 4 * width, 1, s, False))
        convs.append((f"l{li}b{b}.c3", Ho, width, 4 * width, 1, 1, True))
        H, cin = Ho, 4 * width

# stem: conv 7x7 + ReLU + max-pool (csrc/stem.hip)
img = torch.randn(B, 3, 224, 224, device=dev)
wt, bs = torch.randn(147, 64, device=dev) * 0.05, torch.zeros(64, device=dev)
scr = torch.empty(L.ehm_resnet_stem_scratch_bytes(B, 224, 224) // 4, device=dev)
ys = torch.empty(B, 56, 56, 64, device=dev)
for _ in range(2):
    _lib.check(L.ehm_resnet_stem(img.data_ptr(), wt.data_ptr(), bs.data_ptr(), scr.data_ptr(), ys.data_ptr(), B, 224, 224, 0, None))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    _lib.check(L.ehm_resnet_stem(img.data_ptr(), wt.data_ptr(), bs.data_ptr(), scr.data_ptr(), ys.data_ptr(), B, 224, 224, 0, None))
e1.record()
torch.cuda.synchronize()
print(f"stem (pad + conv7x7 + ReLU + max-pool): {e0.elapsed_time(e1) / reps:.3f} ms  (VALU floor 2.1 M cycles/SIMD = 0.87 ms at 2.4 GHz)")
del img, scr, ys

tot_ms = tot_floor = 0.0
rows = []
for name, Hin, Ci, Co, k, s, has_res in convs:
    pad = k // 2
    Ho = (Hin + 2 * pad - k) // s + 1
    K = k * k * Ci
    Co_pad = (Co + 127) // 128 * 128
    w = torch.randn(Co_pad, K, device=dev) * 0.02
    buf = torch.empty(Co_pad, K, device=dev)
    _lib.check(L.ehm_split_pack(w.data_ptr(), buf.data_ptr(), Co_pad, K, K, 1024.0, None))
    bias = torch.zeros(Co, device=dev)
    if X2:
        rin, rout = int(L.ehm_conv_x2_rows(B * Hin * Hin)), int(L.ehm_conv_x2_rows(B * Ho * Ho))
        def x2_random(rows, ch):
            src, dst = torch.randn(rows, ch, device=dev), torch.empty(rows, ch, device=dev)
            src[rows - 1].zero_()
            _lib.check(L.ehm_split_pack(src.data_ptr(), dst.data_ptr(), rows, ch, ch, 1.0, None))
            return dst
        x = x2_random(rin, Ci)
        res = x2_random(rout, Co) if has_res else None
        y = torch.empty(rout, Co, device=dev)
        d = _lib.ConvX2Desc(x.data_ptr(), rin, buf.data_ptr(), bias.data_ptr(), res.data_ptr() if has_res else None, y.data_ptr(),
                            B, Hin, Hin, Ci, Co, k, k, s, pad, 1, 1024.0, None, 0)
        need = int(L.ehm_conv_x2_workspace_bytes(C.byref(d)))           # stream-K scratch (EHM_CONV_NO_STREAMK=1: whole tiles only)
        ws = torch.empty(max(need, 1), dtype=torch.uint8, device=dev)
        if need:
            d.workspace, d.workspace_bytes = ws.data_ptr(), need
        call = lambda: _lib.check(L.ehm_conv_x2(C.byref(d), None))
    else:
        x = torch.randn(B, Hin, Hin, Ci, device=dev)
        res = torch.randn(B, Ho, Ho, Co, device=dev) if has_res else None
        y = torch.empty(B, Ho, Ho, Co, device=dev)
        d = _lib.ConvDesc(x.data_ptr(), buf.data_ptr(), bias.data_ptr(), res.data_ptr() if has_res else None, y.data_ptr(),
                          B, Hin, Hin, Ci, Co, k, k, s, pad, 1, 1024.0)
        call = lambda: _lib.check(L.ehm_conv_nhwc_split(C.byref(d), None))
    for _ in range(2):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    M = B * Ho * Ho
    in_px = B * Hin * Hin if (s == 1 or k == 3) else M          # a strided 1x1 conv touches a quarter of the pixels
    byts = 4.0 * (in_px * Ci + M * Co * (2 if has_res else 1)) + 4.0 * Co_pad * K
    flops = 2.0 * M * K * Co * 3
    t_b, t_f = byts / 5.0e12 * 1e3, flops / 1.0e15 * 1e3
    tot_ms += ms
    tot_floor += max(t_b, t_f)
    rows.append((name, Hin, Ci, Co, k, s, ms, byts / ms / 1e6, flops / ms / 1e9, max(t_b, t_f)))
    del x, w, buf, y, res
print(f"{'conv':10s} {'H':>3s} {'Ci':>5s} {'Co':>5s} k s {'ms':>7s} {'GB/s':>7s} {'TFLOP/s':>8s} {'floor ms':>8s}  (floors: 5.0 TB/s, 1.0 PFLOP/s issued)")
for r in rows:
    print(f"{r[0]:10s} {r[1]:3d} {r[2]:5d} {r[3]:5d} {r[4]} {r[5]} {r[6]:7.3f} {r[7]:7.0f} {r[8]:8.0f} {r[9]:8.3f}")
print(f"total {tot_ms:.2f} ms over {len(rows)} convs; sum of per-conv floors {tot_floor:.2f} ms")
