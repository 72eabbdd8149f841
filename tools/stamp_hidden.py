#!/usr/bin/env python3
"""Per-block phase timing of the register-pipelined split-f16 hidden conv (gcn_f16r.hip) from in-kernel time stamps.
Needs a library built with the stamps compiled in:
    EHM_HIPCC_FLAGS=-DEHM_STAMPS python -c "from egohmr_amd import _lib; _lib.build(force=True)"
    EHM_F16_PIPELINED=2 python tools/stamp_hidden.py f16x3
Prints prologue / K loop / epilogue (loads+fold, barrier, mix+LDS write, barrier, residual+split+store) per block."""
import os, sys, ctypes
import torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egohmr_amd import _lib
from egohmr_amd.factory import build_synthetic_model
prec = sys.argv[1]
dev = torch.device("cuda:0")
model = build_synthetic_model(dev, 0); model.gcn_precision = prec
L = _lib.lib(); h = model.fused_sampler.gcn()
B = int(os.environ.get("EHM_B", "256")); hid = 1024; tile = 192
rows_pad = (2 * B * 24 + tile - 1) // tile * tile
X = torch.randn(rows_pad, hid, device=dev); X2 = torch.empty_like(X); Y1 = torch.empty_like(X); Y2 = torch.empty_like(X)
_lib.check(L.ehm_gcn_pack_activations(X.data_ptr(), X2.data_ptr(), rows_pad, hid, 32, None))
for _ in range(5):
    _lib.check(L.ehm_gcn_hidden_layer(h, 0, X2.data_ptr(), None, Y1.data_ptr(), rows_pad, None))
    _lib.check(L.ehm_gcn_hidden_layer(h, 1, Y1.data_ptr(), X2.data_ptr(), Y2.data_ptr(), rows_pad, None))
torch.cuda.synchronize()
nblk = rows_pad // 192 * 16
dbg = torch.zeros(max(nblk, 1024) * 16, dtype=torch.int64, device=dev)
fn = L.ehm_dbg_set
fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
assert fn(dbg.data_ptr()) == 0
if len(sys.argv) > 2 and sys.argv[2] == "nores":
    _lib.check(L.ehm_gcn_hidden_layer(h, 0, X2.data_ptr(), None, Y2.data_ptr(), rows_pad, None))
else:
    _lib.check(L.ehm_gcn_hidden_layer(h, 1, Y1.data_ptr(), X2.data_ptr(), Y2.data_ptr(), rows_pad, None))
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(-1, 16)[:nblk].astype(np.int64)
t0 = d[:, 0].min()
rt = (d[:, :4] - t0) / 100.0   # us (100 MHz)
print("kernel span us:", rt[:, 3].max())
for name, col in (("start", 0), ("loop0", 1), ("loopend", 2), ("end", 3)):
    v = rt[:, col]; print(f"{name:8s} min {v.min():7.1f} p50 {np.median(v):7.1f} p90 {np.percentile(v,90):7.1f} max {v.max():7.1f}")
pro = rt[:, 1] - rt[:, 0]; loop = rt[:, 2] - rt[:, 1]; epi = rt[:, 3] - rt[:, 2]
for name, v in (("prologue", pro), ("loop", loop), ("epilogue", epi)):
    print(f"{name:8s} us: mean {v.mean():6.2f} p10 {np.percentile(v,10):6.2f} p50 {np.median(v):6.2f} p90 {np.percentile(v,90):6.2f}")
cyc = d[:, 4:8]
for name, a, b in (("prologue", 4, 5), ("loop", 5, 6), ("epilogue", 6, 7)):
    v = d[:, b] - d[:, a]; print(f"{name:8s} cycles: mean {v.mean():9.0f} p50 {np.median(v):9.0f}")
e = (d[:, 8:12] - t0) / 100.0
print("epi: loads+fold %.2f  barrier1 %.2f  mix+ldswrite %.2f  barrier2 %.2f  P2 %.2f (means, us)" % ((e[:,0]-rt[:,2]).mean(), (e[:,1]-e[:,0]).mean(), (e[:,2]-e[:,1]).mean(), (e[:,3]-e[:,2]).mean(), (rt[:,3]-e[:,3]).mean()))
print("epi p50: loads+fold %.2f  barrier1 %.2f  mix+ldswrite %.2f  barrier2 %.2f  P2 %.2f" % (np.median(e[:,0]-rt[:,2]), np.median(e[:,1]-e[:,0]), np.median(e[:,2]-e[:,1]), np.median(e[:,3]-e[:,2]), np.median(rt[:,3]-e[:,3])))
if d[:, 12].any():
    m = (d[:, 12:16] - t0) / 100.0
    print("mix detail (means, us): build+mfma b0 %.2f  scale+ldswrite b0 %.2f  build+mfma b1 %.2f  scale+ldswrite b1 %.2f  rest %.2f" % (
        (m[:, 0] - e[:, 1]).mean(), (m[:, 1] - m[:, 0]).mean(), (m[:, 2] - m[:, 1]).mean(), (m[:, 3] - m[:, 2]).mean(), (e[:, 2] - m[:, 3]).mean()))
first = rt[:, 0] < 5; print("blocks starting <5us:", first.sum(), " second-round start p50:", np.median(rt[~first, 0]) if (~first).any() else None)
