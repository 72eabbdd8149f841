#!/usr/bin/env python3
"""Micro-benchmark of the dominant kernel (one hid->hid Modulated-GCN conv) at the benchmark shape.
Used under rocprofv3 --pmc to collect counters for just this kernel:  python tools/bench_hidden.py f16x3 20"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egohmr_amd import _lib  # noqa: E402
from egohmr_amd.factory import build_synthetic_model  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda:0")
model = build_synthetic_model(dev, 0)
model.gcn_precision = prec
L = _lib.lib()
h = model.fused_sampler.gcn()
hid, tile = model.diffusion_model.hid_dim, L.ehm_gcn_row_tile()
rows_pad = (2 * B * 24 + tile - 1) // tile * tile
X = torch.randn(rows_pad, hid, device=dev) * float(os.environ.get("EHM_X_SCALE", "1"))   # EHM_X_SCALE=0: zero operands (clock / power probe)
if os.environ.get("EHM_X_RELU", "1") != "0":    # default: relu-like activations (half zeros), what the sampler feeds the hidden convs; EHM_X_RELU=0: dense random
    X = torch.relu(X) * 0.5
X2, Y1, Y2 = torch.empty_like(X), torch.empty_like(X), torch.empty_like(X)
if prec != "f32":
    _lib.check(L.ehm_gcn_pack_activations(X.data_ptr(), X2.data_ptr(), rows_pad, hid, L.ehm_gcn_activation_group(h), None))
    X = X2
if os.environ.get("EHM_STACK"):      # time the sampler's own call instead: all hidden convs in one (chained) launch
    import ctypes as C
    bufs = (C.c_void_p * 3)(X.data_ptr(), Y1.data_ptr(), Y2.data_ptr())
    res = C.c_int(0)
    nl = 2 * model.diffusion_model.num_layers
    # the first launches after start-up run at ramping clocks (5 timed launches after 2 warm-up ones read 160 us per split-f16 conv, 400 read 125):
    # warm up for real unless a profiler pass wants few dispatches (EHM_WARMUP=2 in tools/profile_round.sh / pmc_round.sh)
    for _ in range(int(os.environ.get("EHM_WARMUP", "40"))):
        _lib.check(L.ehm_gcn_hidden_stack(h, bufs, rows_pad, C.byref(res), None))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        _lib.check(L.ehm_gcn_hidden_stack(h, bufs, rows_pad, C.byref(res), None))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (nl * reps)
    flops = 2 * B * (24 * 2 * hid * hid + 24 * 24 * hid) * 2.0
    print(f"{prec} B={B} rows_pad={rows_pad} stack of {nl}: {ms * 1e3:.1f} us/conv  {flops / ms / 1e9:.1f} TFLOP/s (algorithmic)")
    sys.exit(0)
for _ in range(2):
    _lib.check(L.ehm_gcn_hidden_layer(h, 0, X.data_ptr(), None, Y1.data_ptr(), rows_pad, None))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(reps):
    _lib.check(L.ehm_gcn_hidden_layer(h, 0, X.data_ptr(), None, Y1.data_ptr(), rows_pad, None))
    _lib.check(L.ehm_gcn_hidden_layer(h, 1, Y1.data_ptr(), X.data_ptr(), Y2.data_ptr(), rows_pad, None))
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / (2 * reps)
flops = 2 * B * (24 * 2 * hid * hid + 24 * 24 * hid) * 2.0
print(f"{prec} B={B} rows_pad={rows_pad}: {ms * 1e3:.1f} us/launch  {flops / ms / 1e9:.1f} TFLOP/s (algorithmic)")
