#!/usr/bin/env python3
"""Timing-only runs of the sampling loop for same-box A/B of library builds (EHM_LIB_PATH), ablation builds included (no finite checks, no
calibration: every step split-f16):  python tools/ab_loop.py [ddpm100|c2_ddim10|c3_guided] [calls]
prints ms per call and the per-launch-class averages of one profiled call (ehm_profile_begin / _end)."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egohmr_amd import _lib, synthetic as syn  # noqa: E402
from egohmr_amd.diffusion import create_gaussian_diffusion  # noqa: E402
from egohmr_amd.factory import batch_to_device, build_synthetic_model  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "ddpm100"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rs = {"ddpm100": "", "c2_ddim10": "ddim10", "c3_guided": ""}[wl]
B, S, guided = (128, 10, True) if wl == "c3_guided" else (256, 1, False)
dev = torch.device("cuda:0")
asset = None
if os.environ.get("EHM_SORT_VERTS"):        # experiment: the synthetic body with its vertices in bone order (real SMPL's vertex numbering is spatially coherent;
    import numpy as np                      # the synthetic asset hangs vertex i on a RANDOM bone, so a wave's 32 vertices gather 32 unrelated transforms)
    asset = syn.make_smpl_asset(0)
    perm = np.argsort(asset["lbs_weights"].argmax(1), kind="stable")
    inv = np.empty_like(perm); inv[perm] = np.arange(len(perm))
    asset = dict(asset, v_template=asset["v_template"][perm], shapedirs=asset["shapedirs"][perm],
                 posedirs=asset["posedirs"].reshape(207, -1, 3)[:, perm].reshape(207, -1).copy(), J_regressor=asset["J_regressor"][:, perm].copy(),
                 lbs_weights=asset["lbs_weights"][perm], faces=inv[asset["faces"]], extra_joints_idxs=inv[asset["extra_joints_idxs"]])
model = build_synthetic_model(dev, 0, sensitive=dict(num_diffusion_timesteps=100), smpl_asset=asset)
model.f16x3_last_steps = None
d = create_gaussian_diffusion(num_diffusion_timesteps=100, timestep_respacing=rs)
T = d.num_timesteps
b = batch_to_device(syn.make_batch(B, 4096, seed=100), dev)
if guided:
    b["scene_pcd_verts_full"][:, : 4096 // 3, 1] = b["smpl_params"]["transl"][:, None, 1] - 0.6
noises = [torch.from_numpy(syn.make_noise_stack(T, B, seed=100 + 1000 * k)).to(dev) for k in range(S)]
fs = model.fused_sampler


def call():
    fs.invalidate()
    fs.run_samples(d, b, noises, ddim=bool(rs), guided=guided, cond_grad_weight=2.0 if guided else 1.0, defer_status=True)


for _ in range(2):
    call()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(calls):
    call()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / calls * 1e3
L = _lib.lib()
L.ehm_profile_begin()
call()
torch.cuda.synchronize()
n = len(_lib.PROF_CLASSES)
ms_arr, cnt = (C.c_double * n)(), (C.c_int64 * n)()
L.ehm_profile_end(ms_arr, cnt, n)
prof = {c: (round(ms_arr[i] / cnt[i] * 1e3, 1), int(cnt[i])) for i, c in enumerate(_lib.PROF_CLASSES) if cnt[i] and ms_arr[i] > 0}
print(f"{os.path.basename(os.environ.get('EHM_LIB_PATH', 'base'))} {wl}: {ms:.2f} ms/call = {B * S / ms * 1e3:.0f} bodies/s  {prof}")
