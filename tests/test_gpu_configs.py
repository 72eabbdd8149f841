"""GPU (MI355X) tests of the BASELINE configurations beyond C1/C2 and of the driver-level pieces:
  C3  B128 x S10 guided DDPM-100 at full size (properties) ;  C4  'ddim50' respacing (reference goldens) and the multi-rank path
  with the real sampler ;  C5  DDPM-1000 + the VolSMPL twin's guidance (reference goldens, default precision schedule and 'f16');
  eval_coll / eval_coll_volsmpl ;  the driver block (S samples -> decode -> metrics -> results pkl) against the oracle pipeline ;
  split-f16 range behaviour."""
import contextlib
import ctypes as C
import os
import pickle
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from egohmr_amd import synthetic as syn

pytestmark = pytest.mark.gpu
VJ_TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def model(dev, synth_weights, smpl_asset):
    from egohmr_amd.factory import build_synthetic_model
    return build_synthetic_model(dev, 0, diffuse_fuse=True, state_dict=synth_weights, smpl_asset=smpl_asset)


@pytest.fixture(scope="module")
def model_vol(dev, synth_weights, smpl_asset):
    from egohmr_amd.factory import build_synthetic_model
    return build_synthetic_model(dev, 0, diffuse_fuse=True, state_dict=synth_weights, smpl_asset=smpl_asset, volsmpl=True)


@contextlib.contextmanager
def precision(m, prec, last_steps="auto"):
    old = (m.gcn_precision, m.f16x3_last_steps)
    m.gcn_precision, m.f16x3_last_steps = prec, last_steps
    try:
        yield m
    finally:
        m.gcn_precision, m.f16x3_last_steps = old


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _check_out(o, g, atol=VJ_TOL, ptol=5e-5):
    c = lambda t: t.detach().cpu().numpy()
    np.testing.assert_allclose(c(o["pred_x_start"]), g["pred_x_start"], atol=ptol)
    np.testing.assert_allclose(c(o["pred_smpl_params"]["betas"]), g["betas"], atol=ptol)
    np.testing.assert_allclose(c(o["pred_smpl_params"]["global_orient"]), g["global_orient"], atol=ptol)
    np.testing.assert_allclose(c(o["pred_smpl_params"]["body_pose"]), g["body_pose"], atol=ptol)
    np.testing.assert_allclose(c(o["pred_vertices"][:, :64]), g["verts_head"], atol=atol)
    np.testing.assert_allclose(c(o["pred_keypoints_3d"]), g["joints"], atol=atol)
    np.testing.assert_allclose(c(o["pred_keypoints_3d_full"]), g["joints_full"], atol=atol)


def _case(g, dev, guided_floor):
    from egohmr_amd.diffusion import create_gaussian_diffusion
    from egohmr_amd.factory import batch_to_device
    B, N, n, rs = int(g["B"]), int(g["N"]), int(g["n"]), str(g["respacing"])
    d = create_gaussian_diffusion(num_diffusion_timesteps=n, timestep_respacing=rs)
    b = batch_to_device(syn.make_batch(B, num_scene_points=N, seed=int(g["batch_seed"])), dev)
    if guided_floor:
        b["scene_pcd_verts_full"][:, : N // 3, 1] = b["smpl_params"]["transl"][:, None, 1] - 0.6
    noise = torch.from_numpy(syn.make_noise_stack(d.num_timesteps, B, seed=int(g["noise_seed"]))).to(dev)
    return d, b, noise, B, rs


# --------------------------------------------------------------------------------------------- C4: 'ddim50'
@pytest.mark.parametrize("name", ["g14_e2e_ddim50_of_100", "g14_e2e_ddim50_of_1000"])
@pytest.mark.parametrize("route", ["fused", "generic"])
def test_c4_ddim50_vs_reference_golden(golden_dir, dev, model, name, route):
    """BASELINE config 4's schedule ('ddim50' of a 100- and of a 1000-step process, respace.py:30-38) through val_losses on the default
    path (f16x3 with the precision schedule) against the reference's own run."""
    g = _load(golden_dir, name)
    d, b, noise, B, rs = _case(g, dev, False)
    assert d.num_timesteps == 50
    with precision(model, "f16x3"):
        d.allow_fused = route == "fused"
        o = d.val_losses(model, b, shape=[B, 144], clip_denoised=False, timestep_respacing=rs, compute_loss=False, noise_stack=noise)
    print(f"[{name}/{route}] max|dverts| = {np.abs(o['pred_vertices'][:, :64].cpu().numpy() - g['verts_head']).max():.3e}")
    _check_out(o, g)


# --------------------------------------------------------------------------------------------- C5: VolSMPL twin, DDPM-1000
@pytest.mark.parametrize("name,route", [("g14_e2e_ddpm50_volsmpl_guided", "fused"), ("g14_e2e_ddpm50_volsmpl_guided", "generic"),
                                        ("g14_e2e_ddpm1000_volsmpl_guided", "fused")])
def test_c5_volsmpl_guided_vs_reference_golden(golden_dir, dev, model_vol, name, route):
    """EgoHMRVolsmpl (models/egohmr/egohmr_volsmpl.py: batched collision loss over ALL scene points, -loss.sum(), w = 30) through the
    reference's own guide_coll / p_sample_with_grad, DDPM-50 and BASELINE config 5's DDPM-1000, plus both collision metrics."""
    from egohmr_amd.model import EgoHMRVolsmpl
    assert isinstance(model_vol, EgoHMRVolsmpl) and model_vol.DEFAULT_COND_GRAD_WEIGHT == 30.0     # test_egohmr_volsmpl.py:62
    g = _load(golden_dir, name)
    d, b, noise, B, rs = _case(g, dev, True)
    with precision(model_vol, "f16x3"):
        d.allow_fused = route == "fused"
        o = d.val_losses(model_vol, b, shape=[B, 144], clip_denoised=False, timestep_respacing=rs, compute_loss=False, noise_stack=noise,
                         cond_fn_with_grad=True, cond_grad_weight=float(g["cond_grad_weight"]))
        coll, coll_v = model_vol.eval_coll(o), model_vol.eval_coll_volsmpl(o)
    print(f"[{name}/{route}] max|dverts| = {np.abs(o['pred_vertices'][:, :64].cpu().numpy() - g['verts_head']).max():.3e}  coll = {coll}")
    _check_out(o, g)
    assert isinstance(coll, list) and len(coll) == B and all(isinstance(c, float) for c in coll)
    # a scene point sitting within float32 noise of the tau shell may flip: allow one point (1/N) per item
    np.testing.assert_allclose(np.array(coll), g["eval_coll"], atol=1.01 / int(g["N"]))
    np.testing.assert_allclose(np.array(coll_v), g["eval_coll_volsmpl"], atol=1.01 / int(g["N"]))
    assert max(coll) > 0


def test_c5_volsmpl_default_weight_30_stays_close(golden_dir, dev, model_vol):
    """The twin's default weight (w = 30, times B through -loss.sum()) drives the build's proxy into a chaotic regime: the fp32 and fp64
    CPU oracles of this very trajectory drift 0.3 apart in x_t over the guided steps (oracle/make_golden.py), so no implementation can
    match the reference's run tightly there.  What can be pinned: the final bodies stay close (the denoiser's x0 is insensitive)."""
    g = _load(golden_dir, "g14_e2e_ddpm50_volsmpl_w30")
    d, b, noise, B, rs = _case(g, dev, True)
    with precision(model_vol, "f16x3"):
        o = d.val_losses(model_vol, b, shape=[B, 144], clip_denoised=False, timestep_respacing=rs, compute_loss=False, noise_stack=noise,
                         cond_fn_with_grad=True, cond_grad_weight=model_vol.DEFAULT_COND_GRAD_WEIGHT)
    j, jr = o["pred_keypoints_3d"][:, :24].cpu().numpy(), g["joints"][:, :24]
    mpjpe_mm = np.linalg.norm((j - j[:, :1]) - (jr - jr[:, :1]), axis=-1).mean() * 1000
    print(f"[volsmpl w=30] MPJPE vs reference = {mpjpe_mm:.3f} mm, max|dx0| = {np.abs(o['pred_x_start'].cpu().numpy() - g['pred_x_start']).max():.3e}")
    assert mpjpe_mm < 10.0 and torch.isfinite(o["pred_vertices"]).all()


def test_c5_fp16_denoiser_ddpm1000_mpjpe_bound(golden_dir, dev, model_vol):
    """BASELINE config 5's 'fp16 denoiser + fp32 LBS' on its own (gcn_precision='f16', no f16x3 steps): NOT parity-grade; pin its
    distance to the reference's DDPM-1000 run as an MPJPE bound."""
    g = _load(golden_dir, "g14_e2e_ddpm1000_volsmpl_guided")
    d, b, noise, B, rs = _case(g, dev, True)
    with precision(model_vol, "f16", None):
        o = d.val_losses(model_vol, b, shape=[B, 144], clip_denoised=False, timestep_respacing=rs, compute_loss=False, noise_stack=noise,
                         cond_fn_with_grad=True, cond_grad_weight=float(g["cond_grad_weight"]))
    j, jr = o["pred_keypoints_3d"][:, :24].cpu().numpy(), g["joints"][:, :24]
    mpjpe_mm = np.linalg.norm((j - j[:, :1]) - (jr - jr[:, :1]), axis=-1).mean() * 1000
    print(f"[f16 DDPM-1000] MPJPE vs reference = {mpjpe_mm:.4f} mm")
    assert mpjpe_mm < 5.0


def test_eval_coll_vs_oracle(dev, model, synth_weights, smpl_asset):
    """EgoHMR.eval_coll (egohmr.py:487-514) on the collision kernel vs the oracle's per-item loop, random bodies over a floor."""
    from egohmr_amd.factory import batch_to_device
    from oracle import model as om
    from oracle.collision import proxy_collision_loss
    B, N = 7, 2048
    bnp = syn.make_batch(B, N, seed=91)
    bnp["scene_pcd_verts_full"][:, : N // 2, 1] = bnp["smpl_params"]["transl"][:, None, 1] - 0.5
    g = np.random.Generator(np.random.PCG64(91))
    x = torch.from_numpy(g.normal(size=(B, 144)).astype(np.float32))
    t = torch.full((B,), 3, dtype=torch.long)
    mean, std = syn.make_body_rep_stats(0)
    ref = om.EgoHMROracle(synth_weights, smpl_asset, mean, std, faithful=False, collision_loss=proxy_collision_loss)
    tb = {k: ({kk: torch.from_numpy(vv) for kk, vv in v.items()} if isinstance(v, dict) else torch.from_numpy(v)) for k, v in bnp.items()}
    tb["x_t"] = x
    ro = ref(tb, t)
    b = batch_to_device(bnp, dev)
    b["x_t"] = x.to(dev)
    with precision(model, "f32"):
        o = model(b, t.to(dev))
    got, want = np.array(model.eval_coll(o)), np.array(ref.eval_coll(ro))
    print("eval_coll:", got, want)
    assert want.max() > 0.01
    np.testing.assert_allclose(got, want, atol=1.01 / N)


# --------------------------------------------------------------------------------------------- C3 at full size
def test_c3_full_size_guided_properties(dev, model):
    """BASELINE config 3 at full size (B128 x S10, guided DDPM-100): too slow for the CPU oracle, so size-independent properties:
    (i) run-to-run determinism - bit-exact unguided, to 2e-5 guided (the gradient scatter onto vertices uses float atomics), (ii) items are independent once the guidance denominator of `-loss.mean()` (egohmr.py:562, the
    GLOBAL batch size) is pinned: a 32-item sub-batch with guide_denom = 128 reproduces its rows, (iii) everything finite, rotations
    orthonormal, (iv) the guidance gradient is zero on joints {0,3,6,9,12..23} (egohmr.py:567) and live on the legs, (v) samples of
    one item differ (S draws) while sharing betas (one conditioning pass)."""
    from egohmr_amd.diffusion import create_gaussian_diffusion
    from egohmr_amd.factory import batch_to_device
    from egohmr_amd.model import GRAD_ZERO_JOINTS
    B, N, S, T = 128, 4096, 10, 100
    d = create_gaussian_diffusion(num_diffusion_timesteps=T, timestep_respacing="")
    bnp = syn.make_batch(B, N, seed=63)
    bnp["scene_pcd_verts_full"][:, : N // 3, 1] = bnp["smpl_params"]["transl"][:, None, 1] - 0.6
    b = batch_to_device(bnp, dev)
    noises = [torch.from_numpy(syn.make_noise_stack(T, B, seed=63 + 100 * s)).to(dev) for s in range(S)]
    fs = model.fused_sampler
    with precision(model, "f16x3"):
        outs = [fs.run(d, b, noises[s], guided=True, cond_grad_weight=2.0)["other_outputs"] for s in range(S)]
        again = fs.run(d, b, noises[3], guided=True, cond_grad_weight=2.0)["other_outputs"]
        for k in ("pred_x_start", "pred_vertices"):
            np.testing.assert_allclose(outs[3][k].cpu().numpy(), again[k].cpu().numpy(), atol=2e-5, err_msg=k)  # (i)
        sub = {k: ({kk: vv[:32] for kk, vv in v.items()} if isinstance(v, dict) else v[:32]) for k, v in b.items()}
        model.guide_denom_override = float(B)
        try:
            part = fs.run(d, sub, noises[0][:, :32].contiguous(), guided=True, cond_grad_weight=2.0)["other_outputs"]
        finally:
            model.guide_denom_override = None
        np.testing.assert_allclose(part["pred_vertices"].cpu().numpy(), outs[0]["pred_vertices"][:32].cpu().numpy(), atol=2e-5)   # (ii)
        unguided = fs.run(d, b, noises[0], guided=False)["other_outputs"]
        assert torch.equal(unguided["pred_vertices"], fs.run(d, b, noises[0], guided=False)["other_outputs"]["pred_vertices"])   # (i)
        # the guidance was live - though faint: `-loss.mean()` divides every item's gradient by B = 128 (the reference's quirk)
        assert float((unguided["pred_vertices"] - outs[0]["pred_vertices"]).norm(dim=-1).max()) > 1e-6
        for o in outs:                                                                                         # (iii)
            assert all(torch.isfinite(o[k]).all() for k in ("pred_x_start", "pred_vertices", "pred_keypoints_3d"))
        R = torch.cat([outs[0]["pred_smpl_params"]["global_orient"], outs[0]["pred_smpl_params"]["body_pose"]], 1)
        assert (R @ R.transpose(-1, -2) - torch.eye(3, device=dev)).abs().max() < 1e-5
        b["x_t"] = noises[0][0] * 0.3
        grad = model.guide_coll(b, outs[0], torch.full((B,), 5, device=dev)).reshape(B, 24, 6)                 # (iv)
        assert float(grad[:, GRAD_ZERO_JOINTS].abs().max()) == 0.0
        assert float(grad[:, [1, 2, 4, 5, 7, 8, 10, 11]].abs().max()) > 0.0
        assert torch.equal(outs[0]["pred_smpl_params"]["betas"], outs[1]["pred_smpl_params"]["betas"])         # (v)
        assert float((outs[0]["pred_x_start"] - outs[1]["pred_x_start"]).abs().max()) > 1e-2


# --------------------------------------------------------------------------------------------- driver block
def test_driver_block_vs_oracle_pipeline(dev, model, synth_weights, smpl_asset, tmp_path):
    """test_egohmr.py:241-266 (S sampling loops, stacked [B,S,...]), :291-318 (decode of the B*S bodies, ground truth), :374-505 (metrics)
    and :672-695 (results pkl) through egohmr_amd.driver.Stage2Driver against oracle/driver.py on B=3, S=2, DDIM-5."""
    from egohmr_amd import io as eio
    from egohmr_amd import smpl as smpl_mod
    from egohmr_amd.diffusion import create_gaussian_diffusion
    from egohmr_amd.driver import Stage2Driver
    from egohmr_amd.factory import batch_to_device
    from oracle import driver as odrv, model as om, schedule as osched
    from oracle.collision import proxy_collision_loss
    from oracle.smpl import SMPLOracle
    B, N, S, n, rs = 3, 2048, 2, 50, "ddim5"
    bnp = syn.make_batch(B, N, seed=17)
    bnp["scene_pcd_verts_full"][:, : N // 3, 1] = bnp["smpl_params"]["transl"][:, None, 1] - 0.6
    gt = syn.make_gt_annotations(B, seed=17)
    bnp["smpl_params"].update({k: gt[k] for k in ("global_orient", "body_pose", "betas")})
    bnp["gender"] = gt["gender"]
    assets = {gname: syn.make_smpl_asset(i) for i, gname in enumerate(("neutral", "male", "female"))}
    assets["neutral"] = smpl_asset
    d = create_gaussian_diffusion(num_diffusion_timesteps=n, timestep_respacing=rs)
    noises = [syn.make_noise_stack(d.num_timesteps, B, seed=17 + 10 * s) for s in range(S)]
    smpls = {k: smpl_mod.create(asset=a, gender=k).to(dev) for k, a in assets.items()}
    with precision(model, "f16x3"):
        drv = Stage2Driver(model, d, smpls["neutral"], smpls["male"], smpls["female"], num_samples=S, timestep_respacing=rs,
                           eval_coll_loss=True, eval_contact_score=True)
        got = drv.step(batch_to_device(bnp, dev), [torch.from_numpy(z).to(dev) for z in noises])
        summary, path = drv.summary(), drv.save(str(tmp_path), "unit", 0)
    mean, std = syn.make_body_rep_stats(0)
    ref_model = om.EgoHMROracle(synth_weights, smpl_asset, mean, std, faithful=False, collision_loss=proxy_collision_loss)
    tb = {k: ({kk: torch.from_numpy(np.asarray(vv)) for kk, vv in v.items()} if isinstance(v, dict) else torch.from_numpy(np.asarray(v)))
          for k, v in bnp.items()}
    want = odrv.run_batch(ref_model, SMPLOracle(assets["neutral"]), SMPLOracle(assets["male"]), SMPLOracle(assets["female"]), tb,
                          osched.make_tables(n, rs), [torch.from_numpy(z) for z in noises], rs, S, eval_coll=True)
    c = lambda t: t.detach().cpu().numpy()
    for k in ("betas", "global_orient", "body_pose"):
        assert got["pred"][k].shape[:2] == (B, S)
        np.testing.assert_allclose(c(got["pred"][k]), want["pred"][k].numpy(), atol=5e-5)
    np.testing.assert_allclose(c(got["decoded"]["vertices"]), want["vertices"].numpy(), atol=VJ_TOL)
    np.testing.assert_allclose(c(got["gt"]["joints"]), want["gt_joints"].numpy(), atol=2e-5)
    np.testing.assert_array_equal(c(got["joint_vis_mask"]), want["joint_vis_mask"].numpy())
    for k in ("g_mpjpe", "mpjpe", "pa_mpjpe", "v2v", "g_mpjpe_vis_sum", "mpjpe_invis_sum", "v2v_vis_sum", "std_joints", "std_joints_vis",
              "apd_joints", "apd_joints_invis", "contact"):
        np.testing.assert_allclose(c(got[k]), want[k], atol=1e-4, rtol=1e-4, equal_nan=True, err_msg=k)
    np.testing.assert_allclose(c(got["coll"]), want["coll"], atol=1.01 / N)
    assert {"G-MPJPE", "MPJPE", "PA-MPJPE", "V2V", "MPJPE-vis", "std-joints", "apd-joints", "contact", "coll"} <= set(summary)
    np.testing.assert_allclose(summary["MPJPE"], 1000 * want["mpjpe"].mean(), rtol=1e-4)
    res = eio.load_results(path)
    with open(path, "rb") as f:
        assert pickle.load(f, encoding="latin1").keys() == res.keys()
    assert res["pred_body_pose_list"].shape == (B, S, 23, 3, 3) and res["collision_ratio_list"].shape == (B, S)
    np.testing.assert_allclose(res["gt_cam_full_list"], bnp["smpl_params"]["transl"])


def test_bench_self_launches_two_ranks_on_this_box(dev):
    """`python bench.py --gpus 2` (no torchrun environment) starts two ranks itself - here sharing this box's one GPU over gloo - and the JSON
    line says so: n_gpus = n_ranks_seen = 2, value = the two ranks' bodies over the max-over-ranks time."""
    import json
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(EGOHMR_DIST_BACKEND="gloo", EGOHMR_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "8", "--scene-points", "512",
                        "--workload", "c1_ddim5", "--cpu-seconds", "0", "--no-legs"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    # stdout = ONE compact JSON line, the LAST one.  Third-party chatter may precede it (gloo's rendezvous lines, and once in round 6 a
    # 78-byte line from the launcher on a fresh box): what the contract fixes is that bench.py itself prints exactly one object and that it comes last
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096, [l[:200] for l in r.stdout.splitlines()]
    assert r.stdout.strip().splitlines()[-1] == lines[0]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["scaling"] == "weak"
    assert abs(out["value"] - 2 * 8 * 1 / (out["ms_per_step"] * 1e-3)) < 1e-3 * out["value"]
    assert len(out["per_rank_bodies_per_s"]) == 2 and all(v > 0 for v in out["per_rank_bodies_per_s"]) and out["all_gather_ms"] > 0
    assert out["roofline"]["avg_launch_ms"] > 0 and out["roofline"]["frac"] > 0 and out["cpu_baseline"] is None
    with open(os.path.join(repo, "bench_detail.json")) as f:                          # the full object of the same run
        detail = json.load(f)
    assert detail["n_ranks_seen"] == 2 and detail["schedule"]["calibrated"] and abs(detail["value"] - out["value"]) < 1e-3 * out["value"]


def test_driver_two_stage_conditions_on_stage1_translation(dev, model, synth_weights, smpl_asset, tmp_path):
    """--two_stage (test_egohmr.py:243-246, :302-303, :691-692): the sampler is conditioned on the stage-1 translation read from the stage-1
    results.pkl (egohmr_amd.io.load_stage1_cam), the ground truth keeps its own, `pred_cam_full_list` lands in the results pkl; against
    oracle/driver.py with two_stage=True, and different from the one-stage run."""
    from egohmr_amd import io as eio
    from egohmr_amd import smpl as smpl_mod
    from egohmr_amd.diffusion import create_gaussian_diffusion
    from egohmr_amd.driver import Stage2Driver
    from egohmr_amd.factory import batch_to_device
    from oracle import driver as odrv, model as om, schedule as osched
    from oracle.smpl import SMPLOracle
    B, N, S, n, rs = 3, 1024, 2, 50, "ddim5"
    bnp = syn.make_batch(B, N, seed=19)
    gt = syn.make_gt_annotations(B, seed=19)
    bnp["smpl_params"].update({k: gt[k] for k in ("global_orient", "body_pose", "betas")})
    bnp["gender"] = gt["gender"]
    stage1 = (bnp["smpl_params"]["transl"] + np.random.Generator(np.random.PCG64(19)).normal(scale=0.08, size=(B, 3))).astype(np.float32)
    pkl_path = tmp_path / "results.pkl"                      # what test_prohmr_scene.py:417-426 writes
    with open(pkl_path, "wb") as f:
        pickle.dump({"pred_cam_full_list": stage1.astype(np.float64), "pred_betas_list": np.zeros((B, 10))}, f, protocol=2)
    bnp["stage1_transl_full"] = eio.load_stage1_cam(str(pkl_path))
    assets = {gname: syn.make_smpl_asset(i) for i, gname in enumerate(("neutral", "male", "female"))}
    assets["neutral"] = smpl_asset
    d = create_gaussian_diffusion(num_diffusion_timesteps=n, timestep_respacing=rs)
    noises = [syn.make_noise_stack(d.num_timesteps, B, seed=19 + 10 * s) for s in range(S)]
    smpls = {k: smpl_mod.create(asset=a, gender=k).to(dev) for k, a in assets.items()}
    with precision(model, "f16x3"):
        drv = Stage2Driver(model, d, smpls["neutral"], smpls["male"], smpls["female"], num_samples=S, timestep_respacing=rs, two_stage=True,
                           eval_contact_score=False)
        got = drv.step(batch_to_device(bnp, dev), [torch.from_numpy(z).to(dev) for z in noises])
        path = drv.save(str(tmp_path), "two_stage", 0)
        one = Stage2Driver(model, d, smpls["neutral"], smpls["male"], smpls["female"], num_samples=S, timestep_respacing=rs, eval_contact_score=False)
        got1 = one.step(batch_to_device({k: v for k, v in bnp.items() if k != "stage1_transl_full"}, dev), [torch.from_numpy(z).to(dev) for z in noises])
        with pytest.raises(KeyError):
            Stage2Driver(model, d, smpls["neutral"], smpls["male"], smpls["female"], num_samples=S, timestep_respacing=rs, two_stage=True).step(
                batch_to_device({k: v for k, v in bnp.items() if k != "stage1_transl_full"}, dev), [torch.from_numpy(z).to(dev) for z in noises])
    mean, std = syn.make_body_rep_stats(0)
    ref_model = om.EgoHMROracle(synth_weights, smpl_asset, mean, std, faithful=False)
    tb = {k: ({kk: torch.from_numpy(np.asarray(vv)) for kk, vv in v.items()} if isinstance(v, dict) else torch.from_numpy(np.asarray(v)))
          for k, v in bnp.items()}
    want = odrv.run_batch(ref_model, SMPLOracle(assets["neutral"]), SMPLOracle(assets["male"]), SMPLOracle(assets["female"]), tb,
                          osched.make_tables(n, rs), [torch.from_numpy(z) for z in noises], rs, S, two_stage=True)
    c = lambda t: t.detach().cpu().numpy()
    np.testing.assert_allclose(c(got["pred"]["body_pose"]), want["pred"]["body_pose"].numpy(), atol=5e-5)
    np.testing.assert_allclose(c(got["decoded"]["joints_full"]), want["joints_full"].numpy(), atol=VJ_TOL)      # pred joints + STAGE-1 translation
    np.testing.assert_allclose(c(got["g_mpjpe"]), want["g_mpjpe"], atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(c(got["gt"]["joints"]), want["gt_joints"].numpy(), atol=2e-5)                     # the ground truth keeps its own
    assert float((got["pred"]["body_pose"] - got1["pred"]["body_pose"]).abs().max()) > 1e-4                      # the conditioning did change
    res = eio.load_results(path)
    np.testing.assert_allclose(res["pred_cam_full_list"], stage1, atol=0)
    np.testing.assert_allclose(res["gt_cam_full_list"], syn.make_batch(B, N, seed=19)["smpl_params"]["transl"])
    assert list(res.keys()).index("pred_cam_full_list") < list(res.keys()).index("gt_cam_full_list")             # the reference's key order (:685-693)


def test_rotmat_to_rot6d_product_vs_reference_golden(golden_dir, dev):
    """utils/geometry.py:69-75 (G2 `rot6d_back`) through the PRODUCT function."""
    from egohmr_amd.geometry import rot6d_to_rotmat, rotmat_to_rot6d
    g = _load(golden_dir, "g2_rot6d")
    R = torch.from_numpy(g["R_diffusion"]).to(dev)
    np.testing.assert_array_equal(rotmat_to_rot6d(R, "diffusion").cpu().numpy(), g["rot6d_back"])
    ok = np.isfinite(g["R_diffusion"]).all(axis=(1, 2))
    back = rot6d_to_rotmat(rotmat_to_rot6d(R, "diffusion"), "diffusion").cpu().numpy()          # a rotation survives the round trip
    np.testing.assert_allclose(back[ok][160:], g["R_diffusion"][ok][160:], atol=1e-4)        # (a few rows have nearly parallel a1, a2)
    with pytest.raises(NotImplementedError):                  # the reference defines the inverse for 'diffusion' only (utils/geometry.py:73-74)
        rotmat_to_rot6d(R, "prohmr")


# --------------------------------------------------------------------------------------------- multi-rank with the real sampler
_RANK_SCRIPT = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, {repo!r})
from egohmr_amd import dist as edist, synthetic as syn
from egohmr_amd.diffusion import create_gaussian_diffusion
from egohmr_amd.factory import batch_to_device, build_synthetic_model
rank, world, _ = edist.init_from_env("gloo")
dev = torch.device("cuda:0")                       # both ranks share the one GPU of the test box
n_items = int(sys.argv[1])
model = build_synthetic_model(dev, 0)
d = create_gaussian_diffusion(num_diffusion_timesteps=50, timestep_respacing="ddim5")
bnp = syn.make_batch(n_items, 1024, seed=5)
noise = syn.make_noise_stack(d.num_timesteps, n_items, seed=5)
items = list(edist.shard_range(n_items, rank, world))
counts = [len(edist.shard_range(n_items, r, world)) for r in range(world)]
sub = batch_to_device({{k: ({{kk: vv[items] for kk, vv in v.items()}} if isinstance(v, dict) else v[items]) for k, v in bnp.items()}}, dev)
edist.agree_schedule(model.fused_sampler, d, sub, ddim=True)      # ONE calibration for the job (rank 0's), not one per shard
o = model.fused_sampler.run(d, sub, torch.from_numpy(noise[:, items]).to(dev), ddim=True)["other_outputs"]
full = edist.gather_packed(edist.pack_params(o["pred_smpl_params"]), counts)
edist.barrier()
if rank == 0:
    np.save(sys.argv[2], full.cpu().numpy())
"""


@pytest.mark.parametrize("n_items", [6, 7])
def test_two_ranks_real_sampler_equals_single_process(dev, model, tmp_path, n_items):
    """BASELINE config 4's data-parallel path with the real FusedSampler: two processes (gloo rendezvous on 127.0.0.1, sharing this
    box's one GPU) each sample their `shard_range` of the items, `gather_packed` assembles the [n,226] rows in item order - which must
    equal the single-process run bit for bit (even and ragged split)."""
    from egohmr_amd import dist as edist
    from egohmr_amd.diffusion import create_gaussian_diffusion
    from egohmr_amd.factory import batch_to_device
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT.format(repo=repo))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = tmp_path / "gathered.npy"
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2",
                   EGOHMR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script), str(n_items), str(out)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        log, _ = p.communicate(timeout=600)
        assert p.returncode == 0, log.decode()[-2000:]
    d = create_gaussian_diffusion(num_diffusion_timesteps=50, timestep_respacing="ddim5")
    bnp = syn.make_batch(n_items, 1024, seed=5)
    noise = torch.from_numpy(syn.make_noise_stack(d.num_timesteps, n_items, seed=5)).to(dev)
    # same precision as the ranks' freshly built models (the module-scoped fixture may have been left in another mode)
    with precision(model, "f16x3"):
        single = {}
        for r in range(2):        # the same shards, one after the other, in this process
            items = list(edist.shard_range(n_items, r, 2))
            sub = batch_to_device({k: ({kk: vv[items] for kk, vv in v.items()} if isinstance(v, dict) else v[items]) for k, v in bnp.items()}, dev)
            if r == 0:
                model.fused_sampler.calibrate_schedule(d, sub, ddim=True, force=True)     # the job's calibration = shard 0's, as agree_schedule does
            o = model.fused_sampler.run(d, sub, noise[:, items].contiguous(), ddim=True)["other_outputs"]
            single[r] = edist.pack_params(o["pred_smpl_params"]).cpu().numpy()
    want = np.concatenate([single[0], single[1]], 0)
    got = np.load(out)
    assert got.shape == (n_items, edist.PACKED_WIDTH)
    np.testing.assert_array_equal(got, want)


# --------------------------------------------------------------------------------------------- split-f16 range behaviour
def _gconv_sd(seed, cin, cout, bn=True, bn_scale=None):
    man = [("l.gconv.W", (2, cin, cout)), ("l.gconv.M", (24, cout)), ("l.gconv.adj2", (24, 24)), ("l.gconv.bias", (cout,))]
    if bn:
        man += [("l.bn.weight", (cout,)), ("l.bn.bias", (cout,)), ("l.bn.running_mean", (cout,)), ("l.bn.running_var", (cout,))]
    sd = {k: torch.from_numpy(v) for k, v in syn.make_state_dict(seed=seed, manifest=man).items()}
    if bn_scale is not None:
        sd["l.bn.weight"][7] = bn_scale           # one channel blown up by its BatchNorm scale
    return sd


@pytest.mark.parametrize("case", ["x1e3", "x1e-3", "bn1e3", "x7e4", "sat2e5", "out1e5"])
def test_split_f16_hidden_conv_range_behaviour(dev, case):
    """The split-f16 ('f16x3') conv outside the O(1) range of the synthetic weights: activations x1e3 and x1e-3, a channel whose
    BatchNorm scale is 1e3, an input element of 7e4 (> f16 max 65504: hi saturates at 65504 and lo carries the remaining 4496, so it is
    still represented) and one of 2e5 (beyond hi + lo: BOTH halves saturate, the element reads as 131008 - the documented limit of the
    format; nothing becomes inf / NaN).  Judge: the fp64 oracle, relative to the output scale.  The clamp is never SILENT: both out-of-range
    cases raise bit 2 of the handle's status word (ehm_gcn_stack_status -> -34 with the remedy in the message, once), the in-range ones do not;
    an output channel driven past 65504 by its BatchNorm scale is caught in the conv's epilogue store."""
    from egohmr_amd import _lib
    from egohmr_amd.model import PRECISIONS
    from oracle import model as om
    from tests.test_gpu_parity import _native_gcn
    L = _lib.lib()
    hid, bodies = 1024, 8
    sds = [_gconv_sd(60, hid, hid, bn_scale={"bn1e3": 1e3, "out1e5": 1e6}.get(case)), _gconv_sd(61, hid, hid)]   # out1e5: one OUTPUT channel past the f16 range
    h, keep = _native_gcn(L, dev, sds[0], sds, _gconv_sd(62, hid, 6, bn=False), hid)
    _lib.check(L.ehm_gcn_set_precision(h, PRECISIONS["f16x3"]))
    g = np.random.Generator(np.random.PCG64(9))
    x = torch.from_numpy(g.normal(size=(bodies, 24, hid)).astype(np.float32))
    x = x * {"x1e3": 1e3, "x1e-3": 1e-3}.get(case, 1.0)
    if case == "x7e4":
        x[2, 5, 100] = 7e4
    if case == "sat2e5":
        x[2, 5, 100] = 2e5
    rows = bodies * 24
    X = x.reshape(rows, hid).to(dev).contiguous()
    T, Y1 = torch.empty_like(X), torch.empty_like(X)
    assert L.ehm_gcn_stack_status(h, None) == 0
    _lib.check(L.ehm_gcn_pack_activations_checked(h, X.data_ptr(), T.data_ptr(), rows, None))
    rc_in = L.ehm_gcn_stack_status(h, None)
    assert rc_in == (-34 if case in ("x7e4", "sat2e5") else 0), (case, rc_in)          # |x| >= 65504 at the format's door: flagged, in range: not
    if rc_in:
        assert b"gcn_precision = 'f32'" in L.ehm_last_error() and L.ehm_gcn_stack_status(h, None) == 0      # reported once, with the remedy
    _lib.check(L.ehm_gcn_hidden_layer(h, 0, T.data_ptr(), None, Y1.data_ptr(), rows, None))
    _lib.check(L.ehm_gcn_unpack_activations(Y1.data_ptr(), T.data_ptr(), rows, hid, 32, None))
    torch.cuda.synchronize()
    y = T.cpu().double()
    rc_out = L.ehm_gcn_stack_status(h, None)                                             # the conv's own stores: flagged iff an output reached the range
    assert (rc_out == -34) == bool(float(y.abs().max()) >= 65504.0), (case, rc_out, float(y.abs().max()))
    xr = x.double().clone()
    if case == "sat2e5":
        xr[2, 5, 100] = 2 * 65504.0                          # hi and lo both saturate at the largest f16
    sd64 = {k.replace("l.", "a."): v.double() for k, v in sds[0].items()}
    r = om._graph_conv(sd64, "a", xr, om.smpl_adjacency().double()).reshape(rows, hid)
    assert torch.isfinite(y).all()
    if case == "out1e5":
        assert rc_out == -34 and float(r.abs().max()) > 65504.0                          # (the case exists to exercise the epilogue guard)
    scale = float(r.abs().max())
    err = float((y - r).abs().max())
    print(f"[{case}] |y|max = {scale:.3e}  max|err| = {err:.3e}  rel = {err / scale:.2e}")
    # outputs above the f16 range are re-split too: hi saturates at 65504 - results beyond ~65536 are clamped by design
    lim = r.abs() < 6.0e4
    # Beyond |x| = 65504 the hi half is saturated and lo is no longer 2^-11 of it, so the dropped lo*lo product shows: the element is
    # still represented, but its products are only good to ~1e-4 relative (documented limit of the format; activations of the
    # denoiser are O(1..100))
    rel = {"x7e4": 1e-4, "sat2e5": 5e-4, "out1e5": 2e-5}.get(case, 3e-6)   # (out1e5: the 1e6 BatchNorm scale multiplies the f32 rounding of its channel)      # (lo / hi = 0.07 resp. 1 instead of 2^-11: the dropped lo*lo term is that much larger)
    assert float((y - r)[lim].abs().max()) < rel * max(1.0, float(r[lim].abs().max()))
    L.ehm_gcn_destroy(h)


def test_sampler_makes_a_clamped_activation_loud_and_can_fall_back_to_f32(dev, smpl_asset):
    """with a checkpoint whose hidden activations leave the f16 range (here: one BatchNorm scale of the first residual block set to 1e6)
    the split-f16 sampling loop no longer returns silently clamped bodies: the call raises EgoHMRRangeError (status bit 2, deferred calls raise at
    check_status()); with EgoHMR.on_saturation = 'f32' it warns, switches the model to float32 activations and returns what a gcn_precision = 'f32' run
    returns, bit for bit.  The untouched checkpoint raises nothing."""
    import warnings
    from egohmr_amd import _lib
    from egohmr_amd.diffusion import create_gaussian_diffusion
    from egohmr_amd.factory import batch_to_device, build_synthetic_model
    B, N = 4, 256
    d = create_gaussian_diffusion(num_diffusion_timesteps=50, timestep_respacing="ddim5")
    batch = batch_to_device(syn.make_batch(B, N, seed=3), dev)
    noise = torch.from_numpy(syn.make_noise_stack(d.num_timesteps, B, seed=3)).to(dev)

    def blown(m):
        with torch.no_grad():
            m.diffusion_model.gconv_layers[0].gconv1.bn.weight[7] = 1e6
        return m

    m = build_synthetic_model(dev, 0, smpl_asset=smpl_asset)
    m.f16x3_last_steps = None
    fs = m.fused_sampler
    fs.run(d, dict(batch), noise, ddim=True)                                   # healthy weights: no flag
    blown(m)
    with pytest.raises(_lib.EgoHMRRangeError, match="gcn_precision = 'f32'"):
        fs.run(d, dict(batch), noise, ddim=True)
    fs.run(d, dict(batch), noise, ddim=True, defer_status=True)               # deferred: the word is looked at later ...
    with pytest.raises(_lib.EgoHMRRangeError):
        fs.check_status()                                                      # ... and is just as loud
    m.on_saturation = "f32"
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = fs.run(d, dict(batch), noise, ddim=True)
    assert any("gcn_precision = 'f32'" in str(x.message) for x in w) and m.gcn_precision == "f32"
    ref = blown(build_synthetic_model(dev, 0, smpl_asset=smpl_asset))
    ref.gcn_precision = "f32"
    want = ref.fused_sampler.run(d, dict(batch), noise, ddim=True)
    assert torch.isfinite(out["other_outputs"]["pred_vertices"]).all()
    assert torch.equal(out["other_outputs"]["pred_vertices"], want["other_outputs"]["pred_vertices"])


def test_forward_under_other_constructor_flags_vs_reference_golden(golden_dir, dev, smpl_asset):
    """The constructor flags outside the shipped test configuration (egohmr.py:31-36): with_bbox_info=False, diffuse_fuse with
    only_mask_img_cond=False (whole-condition second pass), cond_mask_prob > 0 (kept, a no-op in eval) - EgoHMR.forward against the
    reference's own output (g15); with_focal_length=False fails like the reference does (self.with_vfov is never set, :77)."""
    from egohmr_amd.factory import batch_to_device
    from egohmr_amd.model import EgoHMR
    g = _load(golden_dir, "g15_forward_ctor_flags")
    sd = syn.make_state_dict(int(g["weight_seed"]), cam_dim=int(g["cam_dim"]))
    mean, std = syn.make_body_rep_stats(0)
    kw = dict(device=dev, body_rep_mean=mean, body_rep_std=std, with_focal_length=True, with_bbox_info=False, with_cam_center=True,
              scene_feat_dim=512, scene_type="cube", scene_cano=True, cond_mask_prob=0.3, only_mask_img_cond=False, pelvis_vis_loosen=True,
              diffuse_fuse=True, smpl_asset=smpl_asset)
    m = EgoHMR(**kw)
    res = m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=False)
    assert not res.unexpected_keys and all(k.startswith("smpl") for k in res.missing_keys)
    assert m.context_feats_dim == 2048 + 3 + 512 + 128 and m.cond_mask_prob == 0.3
    b = syn.make_batch(3, num_scene_points=int(g["num_scene_points"]), seed=int(g["batch_seed"]))
    b["orig_keypoints_2d"][0, :, 2] = 1.0
    tb = batch_to_device(b, dev)
    tb["x_t"] = torch.from_numpy(g["x_t"]).to(dev)
    for prec in ("f32", "f16x3"):
        with precision(m, prec):
            _check_out(m(tb, torch.from_numpy(g["t"]).to(dev)), g)
    with pytest.raises(AttributeError, match="with_vfov"):
        EgoHMR(**dict(kw, with_focal_length=False))
