#!/usr/bin/env python3
"""Phase timing of skin_mfma_kernel from in-kernel stamps (library built with -DEHM_STAMPS, loaded through EHM_LIB_PATH)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egohmr_amd import _lib  # noqa: E402
from egohmr_amd.factory import build_synthetic_model  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
model = build_synthetic_model(dev, 0)
smpl = model.smpl
g = torch.Generator(device="cpu").manual_seed(0)
betas = torch.randn(B, 10, generator=g).to(dev)
R = torch.linalg.qr(torch.randn(B, 24, 3, 3, generator=g))[0].to(dev)
run = lambda: smpl(betas=betas, body_pose=R[:, 1:], global_orient=R[:, :1], pose2rot=False)
for _ in range(3):
    run()
torch.cuda.synchronize()
L = _lib.lib()
nblk = 4096
dbg = torch.zeros(nblk * 4 * 4, dtype=torch.int64, device=dev)
fn = L.ehm_dbg_set_skin
fn.argtypes = [ctypes.c_void_p]
fn.restype = ctypes.c_int
assert fn(dbg.data_ptr()) == 0
run()
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(nblk * 4, 4)
ok = d[:, 3] > 0
t = d[ok]
print(f"waves stamped {ok.sum()}")
for name, v in (("LDS fill + barrier", t[:, 1] - t[:, 0]), ("blend GEMM (14 k-steps)", t[:, 2] - t[:, 1]), ("skinning + stores", t[:, 3] - t[:, 2]),
                ("wave total", t[:, 3] - t[:, 0])):
    print(f"  {name:26s} mean {v.mean():8.0f}  p10 {np.percentile(v, 10):8.0f}  p50 {np.median(v):8.0f}  p90 {np.percentile(v, 90):8.0f} cycles")
print(f"  kernel span {t[:, 3].max() - t[:, 0].min()} cycles")
