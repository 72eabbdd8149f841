#!/usr/bin/env python3
"""First thing to run when the licensed assets are at hand (they cannot ship; nothing here needs the network):

    python tools/verify_assets.py <smpl_dir | SMPL_NEUTRAL.pkl> <checkpoint.pt> [--stats preprocess_stats.npz] [--mean-params smpl_mean_params.npz]
                                  [--timesteps 50] [--respacing ''] [--batch 32] [--json]

Loads the SMPL model through egohmr_amd.smpl (chumpy-free unpickler) and the checkpoint through egohmr_amd.io.load_checkpoint - the files
the reference reads at models/egohmr/egohmr.py:105-107 and test_egohmr.py:109-111,125-127 - and checks what the synthetic stand-ins of
this repository could never tell:

  1. SMPL asset: 6890 vertices / 13776 faces / 24 joints / 207 pose-corrective directions, J_regressor rows sum to 1 and regress a
     sane skeleton from v_template (parents above children along the spine, left / right symmetric within 1 cm), skinning weights are a
     partition of unity, how many are non-zero per vertex (<= 4: the sparse skinning kernel; more: the dense path), identity pose on
     the GPU reproduces v_template + shapedirs . beta and J_regressor . v_shaped;
  2. checkpoint: load_state_dict(strict=False) leaves NO unexpected key and no missing key outside smpl* / coap*;
  3. precision: on one synthetic batch (the images are noise - conditioning statistics will be off, the arithmetic is what is compared)
     the measured denoiser gain d x0 / d x_t per timestep, FusedSampler.calibrate_schedule's k for these weights, and max vertex distance
     of the 'f16x3', scheduled and 'f16' loops to the 'f32' (f32-input MFMA) loop on the same noise.

Exit code 0 when every hard check passes.  The unit test (tests/test_verify_assets.py) runs it on a synthetic SMPL pkl + state dict.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def check_smpl_asset(asset: dict) -> dict:
    """Host-side structural checks of an SMPL asset dict (egohmr_amd.smpl.load_smpl_asset layout).  -> {name: (ok, detail)}"""
    out = {}
    V = asset["v_template"].shape[0]
    out["vertices_6890"] = (V == 6890, f"{V} vertices")
    out["faces_13776"] = (asset["faces"].shape == (13776, 3), f"faces {asset['faces'].shape}")
    out["faces_index_range"] = (int(asset["faces"].min()) >= 0 and int(asset["faces"].max()) < V, f"[{int(asset['faces'].min())}, {int(asset['faces'].max())}]")
    out["shapedirs_10"] = (asset["shapedirs"].shape == (V, 3, 10), f"{asset['shapedirs'].shape}")
    out["posedirs_207"] = (asset["posedirs"].shape == (207, V * 3), f"{asset['posedirs'].shape}")
    Jr = asset["J_regressor"].astype(np.float64)
    out["j_regressor_shape"] = (Jr.shape == (24, V), f"{Jr.shape}")
    out["j_regressor_rows_sum_to_1"] = (bool(np.allclose(Jr.sum(1), 1.0, atol=1e-4)), f"row sums in [{Jr.sum(1).min():.6f}, {Jr.sum(1).max():.6f}]")
    W = asset["lbs_weights"].astype(np.float64)
    out["skinning_partition_of_unity"] = (bool(np.allclose(W.sum(1), 1.0, atol=1e-4)) and bool((W >= -1e-6).all()), f"row sums in [{W.sum(1).min():.6f}, {W.sum(1).max():.6f}]")
    nnz = (np.abs(W) > 0).sum(1)
    out["skinning_nonzeros_per_vertex"] = (True, f"max {int(nnz.max())} ({'sparse-4 kernel' if nnz.max() <= 4 else 'dense skinning path'}), mean {nnz.mean():.2f}")
    par = np.asarray(asset["parents"]).astype(np.int64).copy()
    par[0] = -1
    out["kinematic_tree"] = (bool((par[1:] < np.arange(1, 24)).all()) and bool((par[1:] >= 0).all()), f"parents {par.tolist()}")
    J = Jr @ asset["v_template"].astype(np.float64)
    bone = np.linalg.norm(J[1:] - J[par[1:]], axis=1)
    out["bone_lengths_sane"] = (bool((bone > 0.01).all() and (bone < 0.6).all()), f"bones {bone.min():.3f} .. {bone.max():.3f} m")
    return out


def main(argv=None) -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("smpl")
    ap.add_argument("checkpoint")
    ap.add_argument("--stats", default=None, help="preprocess_stats.npz (Xmean / Xstd); identity statistics when absent")
    ap.add_argument("--mean-params", default=None, help="data/smpl_mean_params.npz (init betas)")
    ap.add_argument("--timesteps", type=int, default=50, help="num_diffusion_timesteps the checkpoint was trained with (test_egohmr.py:58)")
    ap.add_argument("--respacing", default="")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--scene-points", type=int, default=4096)
    ap.add_argument("--hid", type=int, default=1024)
    ap.add_argument("--blocks", type=int, default=4)
    ap.add_argument("--skip-real-smpl-counts", action="store_true", help="do not insist on 6890 / 13776 (synthetic assets in the unit test keep them anyway)")
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args(argv)

    from egohmr_amd import io as eio
    from egohmr_amd import smpl as smpl_mod
    from egohmr_amd import synthetic as syn
    from egohmr_amd.diffusion import create_gaussian_diffusion
    from egohmr_amd.factory import batch_to_device
    from egohmr_amd.model import EgoHMR

    report, hard_fail = {}, []

    def put(section, name, ok, detail, hard=True):
        report.setdefault(section, {})[name] = {"ok": bool(ok), "detail": detail}
        if hard and not ok:
            hard_fail.append(f"{section}.{name}: {detail}")

    # ---- 1. SMPL asset
    path = smpl_mod.resolve_model_file(a.smpl, "smpl", "neutral") or a.smpl
    asset = smpl_mod.load_smpl_asset(path)
    for k, (ok, det) in check_smpl_asset(asset).items():
        put("smpl", k, ok, det, hard=not (a.skip_real_smpl_counts and k in ("vertices_6890", "faces_13776")))
    assert torch.cuda.is_available(), "the GPU checks need a HIP device; egohmr_amd has no CPU path"
    dev = torch.device("cuda:0")
    body = smpl_mod.create(asset=asset).to(dev)
    g = np.random.Generator(np.random.PCG64(0))
    betas = torch.from_numpy(g.normal(size=(4, 10)).astype(np.float32)).to(dev)
    eye = torch.eye(3, device=dev).expand(4, 24, 3, 3).contiguous()
    o = body(betas=betas, body_pose=eye[:, 1:], global_orient=eye[:, :1], pose2rot=False)
    v_shaped = torch.from_numpy(asset["v_template"]).to(dev)[None] + torch.einsum("vck,bk->bvc", torch.from_numpy(asset["shapedirs"]).to(dev), betas)
    J = torch.einsum("jv,bvc->bjc", torch.from_numpy(asset["J_regressor"]).to(dev), v_shaped)
    ev, ej = float((o.vertices - v_shaped).abs().max()), float((o.joints[:, :24] - J).abs().max())
    put("smpl", "identity_pose_reproduces_v_shaped_on_gpu", ev < 2e-5 and ej < 2e-5, f"max |verts - v_shaped| {ev:.2e}, max |joints - J_regressor v_shaped| {ej:.2e}")

    # ---- 2. checkpoint
    mean, std = eio.load_preprocess_stats(a.stats) if a.stats else syn.make_body_rep_stats(0, identity=True)
    model = EgoHMR(device=dev, body_rep_mean=mean, body_rep_std=std, with_focal_length=True, with_bbox_info=True, with_cam_center=True, scene_feat_dim=512,
                   scene_type="cube", scene_cano=True, cond_mask_prob=0.0, only_mask_img_cond=True, pelvis_vis_loosen=True, diffuse_fuse=True,
                   gcn_hid_dim=a.hid, diffusion_blk=a.blocks, smpl_asset=asset)                      # test_egohmr.py:112-118
    res = eio.load_checkpoint(model, a.checkpoint)
    if a.mean_params:
        model.beta_layer.init_betas.copy_(torch.from_numpy(eio.load_smpl_mean_params(a.mean_params)).to(dev))
    missing = [k for k in res.missing_keys if not k.startswith(("smpl", "coap"))]
    put("checkpoint", "no_unexpected_keys", not [k for k in res.unexpected_keys if not k.startswith(("smpl", "coap", "volume"))],
        f"{len(res.unexpected_keys)} unexpected: {list(res.unexpected_keys)[:6]}")
    put("checkpoint", "no_missing_keys_outside_smpl", not missing, f"{len(missing)} missing: {missing[:6]}")
    model.eval()

    # ---- 3. precision on one batch
    d = create_gaussian_diffusion(num_diffusion_timesteps=a.timesteps, timestep_respacing=a.respacing)
    T, B, ddim = d.num_timesteps, a.batch, bool(a.respacing)
    batch = batch_to_device(syn.make_batch(B, a.scene_points, seed=1), dev)
    noise = torch.from_numpy(syn.make_noise_stack(T, B, seed=1)).to(dev)
    fs = model.fused_sampler
    ts = sorted({d.timestep_map[-1], d.timestep_map[T // 2], d.timestep_map[max(T // 10, 0)], 0}, reverse=True)
    gains = fs.measure_gain(batch, timesteps=ts)
    info = fs.calibrate_schedule(d, batch, ddim=ddim, force=True)
    rows = {}

    saturated = []                       # loops of this checkpoint in which a denoiser activation reached the f16 range (status bit 2: _lib.EgoHMRRangeError)

    def loop(prec, lowprec, name=None):
        model.gcn_precision = prec
        try:
            try:
                return fs.run(d, batch, noise, ddim=ddim, lowprec=lowprec)["other_outputs"]["pred_vertices"].clone()
            except _lib.EgoHMRRangeError:
                saturated.append(name or prec)
                return None
        finally:
            model.gcn_precision = "f16x3"

    from egohmr_amd import _lib
    ref = loop("f32", 0)
    for name, prec, low in (("f16x3", "f16x3", 0), (f"scheduled(k={info['k']})", "f16x3", T - info["k"]), ("f16", "f16", 0)):
        v = loop(prec, low, name)
        if v is None:                                   # clamped activations: the loop's result was refused, not compared
            rows[name] = {"max_vertex_dist_m": float("inf"), "mean_v2v_m": float("inf"), "saturated": True}
            continue
        dv = (v - ref).norm(dim=-1)
        rows[name] = {"max_vertex_dist_m": float(dv.max()), "mean_v2v_m": float(dv.mean())}
    put("precision", "activations_inside_f16_range", not [n for n in saturated if n != "f16"],
        ("no X2 / f16 store of the denoiser clamped" if not saturated else f"clamped in: {saturated}") +
        " (|x| >= 65504 raises status bit 2; remedy: EgoHMR.gcn_precision = 'f32' for this checkpoint, or EgoHMR.on_saturation = 'f32')")
    put("precision", "f16x3_within_1e-4_of_f32", rows["f16x3"]["max_vertex_dist_m"] < 1e-4, json.dumps(rows["f16x3"]))
    put("precision", "scheduled_within_1e-4_of_f32", rows[f"scheduled(k={info['k']})"]["max_vertex_dist_m"] < 1e-4, json.dumps(rows[f"scheduled(k={info['k']})"]))
    put("precision", "f16_reported_only", True, json.dumps(rows["f16"]), hard=False)
    report["precision"]["measured_gain_dx0_dxt"] = gains
    report["precision"]["calibrated_f16x3_last_steps"] = {"k": info["k"], "T": T, "tol_m": info["tol_m"], "trials": info["trials"]}

    if a.json:
        print(json.dumps({"report": report, "hard_failures": hard_fail}))
    else:
        for sec, items in report.items():
            print(f"== {sec}")
            for k, v in items.items():
                if isinstance(v, dict) and "ok" in v:
                    print(f"  [{'ok' if v['ok'] else 'FAIL'}] {k}: {v['detail']}")
                else:
                    print(f"  {k}: {json.dumps(v)}")
        print("hard failures:", hard_fail or "none")
    return 1 if hard_fail else 0


if __name__ == "__main__":
    sys.exit(main())
