#!/usr/bin/env python3
"""How often does the chained hidden-conv kernel actually wait for a producer?  Needs a -DEHM_STAMPS build:
    EHM_HIPCC_FLAGS=-DEHM_STAMPS python -c "from egohmr_amd import _lib; _lib.build(force=True)"; python tools/chain_stats.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egohmr_amd import _lib  # noqa: E402
from egohmr_amd.factory import build_synthetic_model  # noqa: E402

dev = torch.device("cuda:0")
model = build_synthetic_model(dev, 0)
L = _lib.lib()
h = model.fused_sampler.gcn()
B, hid, tile = 256, model.diffusion_model.hid_dim, L.ehm_gcn_row_tile()
rows_pad = (2 * B * 24 + tile - 1) // tile * tile
x0 = torch.relu(torch.randn(rows_pad, hid, device=dev)) * 0.5
X = [torch.empty_like(x0) for _ in range(3)]
_lib.check(L.ehm_gcn_pack_activations(x0.data_ptr(), X[0].data_ptr(), rows_pad, hid, 32, None))
X0 = X[0].clone()
bufs = (C.c_void_p * 3)(*[t.data_ptr() for t in X])
res = C.c_int(0)
fn = L.ehm_dbg_chain_stats
fn.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
fn.restype = C.c_int
for rep in range(4):
    X[0].copy_(X0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(L.ehm_gcn_hidden_stack(h, bufs, rows_pad, C.byref(res), None))
    e1.record()
    torch.cuda.synchronize()
    st = (C.c_uint * 3)()
    if fn(h, st) != 0:
        st[0] = st[1] = st[2] = 0
    print(f"rep {rep}: {e0.elapsed_time(e1) * 1e3:.0f} us for 8 convs ({e0.elapsed_time(e1) * 125:.1f} us/conv); err={st[0]} spins={st[1]} waits_that_spun={st[2]} of {8 * 64 * 16 - 1024} dependent tiles")
