#!/usr/bin/env python3
"""Micro-benchmark of the hoisted input conv (gcn_input_kernel) at the benchmark shape: python tools/bench_input.py [f16x3|f16|f32] [reps] [B]
EHM_LIB_PATH selects another build (timing-only ablations: EHM_HIPCC_FLAGS=-DEHM_ABL_IN_...)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egohmr_amd import _lib  # noqa: E402
from egohmr_amd.factory import build_synthetic_model  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda:0")
model = build_synthetic_model(dev, 0)
model.gcn_precision = prec
L = _lib.lib()
fs = model.fused_sampler
h = fs.gcn()
hid = model.diffusion_model.hid_dim
rows = 2 * B * 24
h_img, h_oth = torch.randn(B, 2, hid, device=dev), torch.randn(B, 2, hid, device=dev)
vis = (torch.rand(B, 24, device=dev) < 0.6).to(torch.uint8)
x = torch.randn(B, 144, device=dev)
tv = fs.timestep_vectors([5])
Y = torch.empty(rows + 192, hid, device=dev)
Wx = fs._folded.Wx if hasattr(fs._folded, "Wx") else None
args = (h, h_img.data_ptr(), h_oth.data_ptr(), vis.data_ptr(), x.data_ptr(), Wx.data_ptr(), tv.data_ptr(), Y.data_ptr(), B, 2, None)
for _ in range(20):
    _lib.check(L.ehm_gcn_input_layer(*args))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    _lib.check(L.ehm_gcn_input_layer(*args))
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
print(f"{os.environ.get('EHM_LIB_PATH', 'shipped'):40s} {prec} B={B}: {us:.1f} us per launch, {rows * hid * 4 / us / 1e6:.2f} TB/s of rows written")
