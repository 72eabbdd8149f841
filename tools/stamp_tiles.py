#!/usr/bin/env python3
"""Per-tile phase timing of the hidden-conv tile engine (csrc/gcn_tile.hip) from in-kernel s_memtime stamps.
Needs a library built with the stamps compiled in:
    EHM_HIPCC_FLAGS=-DEHM_STAMPS python -c "from egohmr_amd import _lib; _lib.build(force=True)"
    python tools/stamp_tiles.py f16|f16x3 [chain|layer]
Slots per tile: 0 head (before vmcnt(0)+barrier), 5 after the head barrier, 1 first fragments requested, 2 K loop done (last MFMAs
issued), 3 next tile's DMA issued, 4 epilogue done.  Cycles are shader-clock ticks of the block's first wave."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egohmr_amd import _lib  # noqa: E402
from egohmr_amd.factory import build_synthetic_model  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
mode = sys.argv[2] if len(sys.argv) > 2 else "chain"
dev = torch.device("cuda:0")
model = build_synthetic_model(dev, 0)
model.gcn_precision = prec
L = _lib.lib()
h = model.fused_sampler.gcn()
B = int(os.environ.get("EHM_B", "256"))
hid, tile = 1024, 192
rows_pad = (2 * B * 24 + tile - 1) // tile * tile
X = torch.relu(torch.randn(rows_pad, hid, device=dev)) * 0.5
X2, Y1, Y2 = torch.empty_like(X), torch.empty_like(X), torch.empty_like(X)
_lib.check(L.ehm_gcn_pack_activations(X.data_ptr(), X2.data_ptr(), rows_pad, hid, L.ehm_gcn_activation_group(h), None))
bufs = (ctypes.c_void_p * 3)(X2.data_ptr(), Y1.data_ptr(), Y2.data_ptr())
res = ctypes.c_int(0)


def run():
    if mode == "chain":
        _lib.check(L.ehm_gcn_hidden_stack(h, bufs, rows_pad, ctypes.byref(res), None))
    else:
        _lib.check(L.ehm_gcn_hidden_layer(h, 1, Y1.data_ptr(), X2.data_ptr(), Y2.data_ptr(), rows_pad, None))


for _ in range(3):
    run()
torch.cuda.synchronize()
nblk = int(os.environ.get("EHM_CHAIN_BLOCKS", "512")) if mode == "chain" else rows_pad // 192 * 16
dbg = torch.zeros(nblk * 64 * 8, dtype=torch.int64, device=dev)
fn = L.ehm_dbg_set
fn.argtypes = [ctypes.c_void_p]
fn.restype = ctypes.c_int
assert fn(dbg.data_ptr()) == 0
run()
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(nblk, 64, 8).astype(np.int64)
ok = d[:, :, 4] > 0
print(f"{prec} {mode}: tiles stamped {ok.sum()} (ring of 64 per block; {nblk} blocks)")
t = d[ok]
seg = {"head wait (prev stores + DMA landing)": t[:, 5] - t[:, 0], "consts + first frags": t[:, 1] - t[:, 5], "K loop": t[:, 2] - t[:, 1],
       "next-tile DMA issue (+slot barrier)": t[:, 3] - t[:, 2], "epilogue": t[:, 4] - t[:, 3],
       "  epi: tables + fold": t[:, 6] - t[:, 3], "  epi: adjacency mix": t[:, 7] - t[:, 6], "  epi: LDS turn + residual + stores": t[:, 4] - t[:, 7],
       "tile total": t[:, 4] - t[:, 0]}
for k, v in seg.items():
    print(f"  {k:40s} mean {v.mean():9.0f}  p10 {np.percentile(v, 10):9.0f}  p50 {np.median(v):9.0f}  p90 {np.percentile(v, 90):9.0f} cycles")
if mode == "chain":
    # K loop / tile total by the block's tile number (16 tiles per block at B = 256: layer = i // 2, i % 2 = first / second tile of the layer)
    kl, tt = d[:, :, 2] - d[:, :, 1], d[:, :, 4] - d[:, :, 0]
    n_i = int(ok.sum(1).max())
    print("  per tile number of the block:  " + "  ".join(f"{i}:{np.median(kl[:, i][ok[:, i]]) / 1e3:.0f}k/{np.median(tt[:, i][ok[:, i]]) / 1e3:.0f}k" for i in range(n_i) if ok[:, i].any()))
    # gap between a tile's end and the next tile's head on the same block
    gaps = []
    for b in range(nblk):
        idx = np.where(ok[b])[0]
        for i in idx[1:]:
            if ok[b, i - 1]:
                gaps.append(d[b, i, 0] - d[b, i - 1, 4])
    gaps = np.array(gaps)
    if gaps.size:
        print(f"  {'between tiles (publish / late deps)':40s} mean {gaps.mean():9.0f}  p50 {np.median(gaps):9.0f}  p90 {np.percentile(gaps, 90):9.0f}")
    span = d[:, :, 4].max() - d[:, :, 0][ok].min()
    print(f"  kernel span {span} cycles; tiles per block mean {ok.sum(1).mean():.1f}")
