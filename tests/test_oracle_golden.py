"""CPU: the oracle restatement against golden vectors produced by the reference itself
(oracle/make_golden.py).  Pins schedule, geometry, GCN, embedders, encoders, single sampler steps,
EgoHMR.forward and the end-to-end DDIM/DDPM loops (guided plumbing included)."""
import os

import numpy as np
import pytest
import torch

from egohmr_amd import synthetic as syn
from oracle import geometry as geo
from oracle import model as om
from oracle import sampler, schedule
from oracle.collision import proxy_collision_loss


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _tt(b):
    return {k: (_tt(v) if isinstance(v, dict) else torch.from_numpy(np.asarray(v))) for k, v in b.items()}


@pytest.mark.parametrize("n,rs", [(50, ""), (50, "ddim5"), (50, "ddim10"), (100, ""), (100, "ddim10"), (100, "ddim50"),
                                  (1000, ""), (1000, "ddim10"), (1000, "ddim50")])
def test_g1_schedule_tables(golden_dir, n, rs):
    g = _load(golden_dir, "g1_schedules")
    t = schedule.make_tables(n, rs)
    tag = f"n{n}_{rs or 'ddpm'}"
    for f in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
              "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
              "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"):
        np.testing.assert_array_equal(getattr(t, f), g[f"{tag}__{f}"], err_msg=f)   # float64, bit-exact
    assert t.timestep_map == list(g[f"{tag}__timestep_map"])


def test_bad_respacing_raises():
    with pytest.raises(ValueError):
        schedule.make_tables(50, "ddim49")


def test_g2_rot6d(golden_dir):
    g = _load(golden_dir, "g2_rot6d")
    x = torch.from_numpy(g["x"])
    np.testing.assert_allclose(geo.rot6d_to_rotmat(x, "diffusion").numpy(), g["R_diffusion"], atol=2e-6)
    np.testing.assert_allclose(geo.rot6d_to_rotmat(x, "prohmr").numpy(), g["R_prohmr"], atol=2e-6)
    np.testing.assert_array_equal(geo.rotmat_to_rot6d(torch.from_numpy(g["R_diffusion"])).numpy(), g["rot6d_back"])
    np.testing.assert_allclose(geo.aa_to_rotmat(torch.from_numpy(g["aa"])).numpy(), g["R_from_aa"], atol=1e-6)


def test_g3_rotmat_to_angle_axis(golden_dir):
    g = _load(golden_dir, "g3_rotmat_to_aa")
    np.testing.assert_allclose(geo.rotation_matrix_to_angle_axis(torch.from_numpy(g["R"])).numpy(), g["aa"], atol=1e-5)


def test_g4_gcn(golden_dir):
    g = _load(golden_dir, "g4_gcn_tiny")
    sd = {"diffusion_model." + k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w__")}
    y = om.modulated_gcn(sd, torch.from_numpy(g["x"]), torch.from_numpy(g["adj"]), num_blocks=1)
    np.testing.assert_allclose(y.numpy(), g["y"], atol=1e-5)
    np.testing.assert_allclose(om.smpl_adjacency().numpy(), g["adj"], atol=0)
    g = _load(golden_dir, "g4_gconv_1024")
    man = [("gconv.W", (2, 1024, 1024)), ("gconv.M", (24, 1024)), ("gconv.adj2", (24, 24)), ("gconv.bias", (1024,))]
    sd = {k: torch.from_numpy(v) for k, v in syn.make_state_dict(seed=int(g["weight_seed"]), manifest=man).items()}
    y = om.modulated_graph_conv(sd, "gconv", torch.from_numpy(g["x"]), om.smpl_adjacency())
    np.testing.assert_allclose(y.numpy(), g["y"], atol=2e-5)


def test_g5_g6_embedders_encoders(golden_dir, synth_weights):
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth_weights.items()}
    g = _load(golden_dir, "g5_timestep_embed")
    np.testing.assert_allclose(om.timestep_embedding(sd, torch.from_numpy(g["t"])).numpy(), g["emb"], atol=1e-6)
    g = _load(golden_dir, "g6_pointnet")
    np.testing.assert_allclose(om.resnet_pointnet(sd, torch.from_numpy(g["pts"])).numpy(), g["feat"], atol=1e-5)
    g = _load(golden_dir, "g6_resnet50")
    rng = np.random.Generator(np.random.PCG64(int(g["img_seed"])))
    rng.uniform(-1, 1, size=(2, 257, 3))          # same stream position as the generator script
    img = rng.normal(size=(2, 3, 224, 224)).astype(np.float32)
    np.testing.assert_allclose(om.resnet50(sd, torch.from_numpy(img)).numpy(), g["feat"], atol=2e-5)


def test_g7_single_steps(golden_dir):
    g = _load(golden_dir, "g7_single_steps")

    class Dummy:
        def __init__(self, x0):
            self.x0 = x0

        def validation_setup(self):
            pass

        def __call__(self, batch, t):
            self.t = t
            return {"pred_x_start": self.x0}

    for n, rs, idx in [(50, "", 49), (50, "", 7), (50, "", 0), (100, "ddim10", 9), (100, "ddim10", 3), (100, "ddim10", 0),
                       (100, "ddim50", 49), (100, "ddim50", 17), (100, "ddim50", 0), (1000, "ddim50", 49), (1000, "ddim50", 1),
                       (1000, "", 999), (1000, "", 500), (1000, "", 3), (1000, "", 0)]:      # + BASELINE config 4 / 5 schedules
        tag = f"n{n}_{rs or 'ddpm'}_i{idx}"
        full = schedule.make_tables(n, rs)
        # run exactly one step: tables truncated so that the loop's single index is ``idx``
        one = schedule.Tables(**{k: (v[idx:idx + 1] if isinstance(v, np.ndarray) else [v[idx]]) for k, v in full.__dict__.items()})
        x, x0, eps = (torch.from_numpy(g[f"{tag}__{k}"]) for k in ("x", "x0", "eps"))
        m = Dummy(x0)
        noise = torch.stack([x, eps])
        fn = sampler.ddim_sample_loop if rs else sampler.p_sample_loop
        if idx == 0:
            o = fn(m, {}, one, noise)
        else:  # loop treats its last index as t == 0 -> emulate "not last" by a 2-entry table whose 2nd step we take
            two = schedule.Tables(**{k: (np.concatenate([v[:1], v[idx:idx + 1]]) if isinstance(v, np.ndarray) else [v[0], v[idx]])
                                     for k, v in full.__dict__.items()})
            tr = []
            fn(m, {}, two, torch.stack([x, eps, eps]), trace=tr)
            o = {"sample": tr[0][0]}
        np.testing.assert_allclose(o["sample"].numpy(), g[f"{tag}__sample"], atol=1e-6, err_msg=tag)


def _model(synth_weights, smpl_asset, **kw):
    mean, std = syn.make_body_rep_stats(0)
    return om.EgoHMROracle(synth_weights, smpl_asset, mean, std, **kw)


def _check_out(o, g, prefix="", atol=2e-5):
    np.testing.assert_allclose(o["pred_x_start"].numpy(), g[prefix + "pred_x_start"], atol=atol)
    np.testing.assert_allclose(o["pred_smpl_params"]["betas"].numpy(), g[prefix + "betas"], atol=atol)
    np.testing.assert_allclose(o["pred_smpl_params"]["global_orient"].numpy(), g[prefix + "global_orient"], atol=atol)
    np.testing.assert_allclose(o["pred_smpl_params"]["body_pose"].numpy(), g[prefix + "body_pose"], atol=atol)
    np.testing.assert_allclose(o["pred_vertices"][:, :64].numpy(), g[prefix + "verts_head"], atol=atol)
    np.testing.assert_allclose(o["pred_vertices"].double().sum(1).numpy(), g[prefix + "verts_sum"], atol=2e-2)
    np.testing.assert_allclose(o["pred_keypoints_3d"].numpy(), g[prefix + "joints"], atol=atol)
    np.testing.assert_allclose(o["pred_keypoints_3d_full"].numpy(), g[prefix + "joints_full"], atol=atol)
    np.testing.assert_allclose(o["pred_keypoints_2d_full"].numpy(), g[prefix + "kp2d_full"], atol=atol)


def test_g10_forward(golden_dir, synth_weights, smpl_asset):
    g = _load(golden_dir, "g10_forward")
    b = syn.make_batch(3, num_scene_points=int(g["num_scene_points"]), seed=int(g["batch_seed"]))
    b["orig_keypoints_2d"][0, :, 2] = 1.0
    b["orig_keypoints_2d"][1, :, 2] = 0.0
    for tag, fuse in (("fuse__", True), ("nofuse__", False)):
        m = _model(synth_weights, smpl_asset, diffuse_fuse=fuse)
        tb = _tt(b)
        tb["x_t"] = torch.from_numpy(g["x_t"])
        o = m(tb, torch.from_numpy(g["t"]))
        _check_out(o, g, tag)
        np.testing.assert_array_equal(tb["vis_mask_smpl"].numpy(), g[tag + "vis_mask_smpl"])


@pytest.mark.parametrize("name", ["g8_e2e_ddim5", "g9_e2e_ddpm50", "g9_e2e_ddpm50_guided", "g12_e2e_ddim10_guided"])
@pytest.mark.parametrize("faithful", [True, False])
def test_g8_g9_end_to_end(golden_dir, synth_weights, smpl_asset, name, faithful):
    if faithful and name != "g8_e2e_ddim5":
        pytest.skip("reference-faithful mode (encoders every step) checked on the short loop only")
    g = _load(golden_dir, name)
    B, N, n, rs = int(g["B"]), int(g["N"]), int(g["n"]), str(g["respacing"])
    guided = bool(g["guided"])
    b = _tt(syn.make_batch(B, num_scene_points=N, seed=int(g["batch_seed"])))
    if guided:
        b["scene_pcd_verts_full"][:, : N // 3, 1] = b["smpl_params"]["transl"][:, None, 1] - 0.6
    tab = schedule.make_tables(n, rs)
    noise = torch.from_numpy(syn.make_noise_stack(tab.num_timesteps, B, seed=int(g["noise_seed"])))
    m = _model(synth_weights, smpl_asset, faithful=faithful, collision_loss=proxy_collision_loss)
    tr = []
    o = sampler.val_losses(m, b, tab, noise, rs, cond_fn_with_grad=guided, cond_grad_weight=float(g["cond_grad_weight"]), trace=tr)
    xs = np.stack([noise[0].numpy()] + [t[0].numpy() for t in tr[:-1]])
    np.testing.assert_allclose(xs, g["x_t_trace"], atol=2e-5)
    _check_out(o, g)
    if name == "g12_e2e_ddim10_guided":       # the guidance must have been live: the unguided trajectory differs on the last steps
        tr0 = []
        sampler.val_losses(m, b, tab, noise, rs, cond_fn_with_grad=False, trace=tr0)
        assert float((tr0[-1][0] - tr[-1][0]).abs().max()) > 2e-5   # (small: scale 1.0 x sqrt(1 - abar) on the last steps)
        assert float((tr0[5][0] - tr[5][0]).abs().max()) == 0.0   # ... and only there (t <= 3 of 10)


def test_g13_gcn_with_non_local_block(golden_dir):
    """oracle ModulatedGCN + NONLocalBlock2D against the reference module (nonlocal_layer=True, modulated_gcn.py:93-110)."""
    g = _load(golden_dir, "g13_gcn_nonlocal")
    hid, in_dim = int(g["hid"]), int(g["in_dim"])
    man = [(n, sh) for n, sh in syn.egohmr_manifest(hid_dim=hid, num_blocks=1, with_backbone=False, nonlocal_layer=True)
           if n.startswith("diffusion_model.")]
    man = [(n, ((2, in_dim, hid) if n == "diffusion_model.gconv_input.0.gconv.W" else sh)) for n, sh in man]
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in syn.make_state_dict(seed=int(g["weight_seed"]), manifest=man).items()}
    x = torch.from_numpy(g["x"])
    y = om.modulated_gcn(sd, x, om.smpl_adjacency(), num_blocks=1, nonlocal_layer=True)
    np.testing.assert_allclose(y.numpy(), g["y"], atol=1e-5)
    y0 = om.modulated_gcn(sd, x, om.smpl_adjacency(), num_blocks=1, nonlocal_layer=False)
    assert float((y - y0).abs().max()) > 1e-2          # the block is not an identity with these weights


@pytest.mark.parametrize("name", ["g14_e2e_ddim50_of_100", "g14_e2e_ddpm50_volsmpl_guided"])
def test_g14_c4_schedule_and_volsmpl_twin(golden_dir, synth_weights, smpl_asset, name):
    """BASELINE config 4's 'ddim50' respacing end to end, and the VolSMPL twin (models/egohmr/egohmr_volsmpl.py:582-629: batched
    collision loss over ALL scene points, -loss.sum(), w = 30) with both collision metrics (egohmr.py:487-514,
    egohmr_volsmpl.py:548-579), against the reference's own run (oracle/make_golden.py g14_*).  The 1000-step goldens of the set
    are checked on the GPU only (tests/test_gpu_configs.py): a thousand CPU evaluations do not belong in this suite."""
    g = _load(golden_dir, name)
    B, N, n, rs = int(g["B"]), int(g["N"]), int(g["n"]), str(g["respacing"])
    guided = bool(g["guided"])
    b = _tt(syn.make_batch(B, num_scene_points=N, seed=int(g["batch_seed"])))
    if guided:
        b["scene_pcd_verts_full"][:, : N // 3, 1] = b["smpl_params"]["transl"][:, None, 1] - 0.6
    tab = schedule.make_tables(n, rs)
    noise = torch.from_numpy(syn.make_noise_stack(tab.num_timesteps, B, seed=int(g["noise_seed"])))
    m = _model(synth_weights, smpl_asset, faithful=False, collision_loss=proxy_collision_loss)
    o = sampler.val_losses(m, b, tab, noise, rs, cond_fn_with_grad=guided, cond_grad_weight=float(g["cond_grad_weight"]),
                           guide_reduction="sum" if guided else "mean", guide_all_points=guided)
    _check_out(o, g)
    if guided:
        np.testing.assert_allclose(np.array(m.eval_coll(o)), g["eval_coll"], atol=1e-12)
        np.testing.assert_allclose(np.array(m.eval_coll(o)), g["eval_coll_volsmpl"], atol=1e-12)
        assert g["eval_coll"].max() > 0                       # the floor does cut through the bodies of this fixture
        o_mean = sampler.val_losses(m, b, tab, noise, rs, cond_fn_with_grad=True, cond_grad_weight=float(g["cond_grad_weight"]))
        assert float((o_mean["pred_x_start"] - o["pred_x_start"]).abs().max()) > 3e-5     # ... and the COAP-style variant (mean over B, bbox) differs


def test_g15_forward_under_other_constructor_flags(golden_dir, smpl_asset):
    """with_bbox_info=False (3 camera features instead of 6), diffuse_fuse with only_mask_img_cond=False (the second pass masks the WHOLE
    condition, mask_cond egohmr.py:156-157), cond_mask_prob > 0 (training-only): the oracle vs the reference's own forward."""
    g = _load(golden_dir, "g15_forward_ctor_flags")
    sd = syn.make_state_dict(int(g["weight_seed"]), cam_dim=int(g["cam_dim"]))
    mean, std = syn.make_body_rep_stats(0)
    m = om.EgoHMROracle(sd, smpl_asset, mean, std, diffuse_fuse=True, with_bbox_info=False, with_cam_center=True, only_mask_img_cond=False)
    b = syn.make_batch(3, num_scene_points=int(g["num_scene_points"]), seed=int(g["batch_seed"]))
    b["orig_keypoints_2d"][0, :, 2] = 1.0
    tb = _tt(b)
    tb["x_t"] = torch.from_numpy(g["x_t"])
    _check_out(m(tb, torch.from_numpy(g["t"])), g)


# --------------------------------------------------------------------------------------------- x_t-sensitive synthetic denoiser (G16)
@pytest.fixture(scope="module")
def sensitive_weights():
    return syn.make_sensitive_state_dict(0, 100)


def test_g16_forward_sensitive(golden_dir, sensitive_weights, smpl_asset):
    """oracle forward with the x_t-sensitive weights at high / mid / low noise vs the reference's own forward."""
    g = _load(golden_dir, "g16_forward_sensitive")
    b = syn.make_batch(3, num_scene_points=int(g["num_scene_points"]), seed=int(g["batch_seed"]))
    b["orig_keypoints_2d"][0, :, 2] = 1.0
    m = _model(sensitive_weights, smpl_asset, faithful=False)
    for t in g["ts"]:
        tb = _tt(b)
        tb["x_t"] = torch.from_numpy(g["x_t"])
        _check_out(m(tb, torch.full((3,), int(t))), g, f"t{int(t)}__", atol=5e-5)


@pytest.mark.parametrize("name", ["g16_e2e_ddim10_sensitive", "g16_e2e_ddpm100_sensitive"])
def test_g16_end_to_end_sensitive(golden_dir, sensitive_weights, smpl_asset, name):
    g = _load(golden_dir, name)
    B, N, n, rs = int(g["B"]), int(g["N"]), int(g["n"]), str(g["respacing"])
    b = _tt(syn.make_batch(B, num_scene_points=N, seed=int(g["batch_seed"])))
    tab = schedule.make_tables(n, rs)
    noise = torch.from_numpy(syn.make_noise_stack(tab.num_timesteps, B, seed=int(g["noise_seed"])))
    m = _model(sensitive_weights, smpl_asset, faithful=False)
    tr = []
    o = sampler.val_losses(m, b, tab, noise, rs, trace=tr)
    xs = np.stack([noise[0].numpy()] + [t[0].numpy() for t in tr[:-1]])
    np.testing.assert_allclose(xs, g["x_t_trace"], atol=1e-4)
    _check_out(o, g, atol=1e-4)


def test_g18_rows_of_the_full_size_golden(golden_dir, sensitive_weights, smpl_asset):
    """The oracle against the reference's FULL-SIZE run of BASELINE config 2 (golden G18: 256 items, DDIM-10 of 100, 4096 scene points): items are independent
    (BatchNorm in eval mode, per-item conditioning), so the oracle on three items of that batch - the first two and the last - must reproduce their rows."""
    g = _load(golden_dir, "g18_c2_ddim10_b256_sensitive")
    B, N, n, rs = int(g["B"]), int(g["N"]), int(g["n"]), str(g["respacing"])
    rows = [0, 1, B - 1]
    full = syn.make_batch(B, num_scene_points=N, seed=int(g["batch_seed"]))
    sub = {k: ({kk: vv[rows] for kk, vv in v.items()} if isinstance(v, dict) else v[rows]) for k, v in full.items()}
    tab = schedule.make_tables(n, rs)
    noise = torch.from_numpy(syn.make_noise_stack(tab.num_timesteps, B, seed=int(g["noise_seed"]))[:, rows])
    m = _model(sensitive_weights, smpl_asset, faithful=False)
    o = sampler.val_losses(m, _tt(sub), tab, noise, rs)
    gs = {k: (g[k][rows] if getattr(g[k], "ndim", 0) >= 1 and g[k].shape[0] == B else g[k]) for k in g.files}
    _check_out(o, gs, atol=1e-4)


def test_g17_end_to_end_gain03(golden_dir, smpl_asset):
    """the oracle on the partially sensitive weights (low-noise gain 0.3) vs the reference's own DDPM-100 on them (golden G17: the
    reference gate of a MIXED plain-f16 / split-f16 schedule, tests/test_gpu_schedule.py)."""
    g = _load(golden_dir, "g17_e2e_ddpm100_gain03")
    B, N, n, rs = int(g["B"]), int(g["N"]), int(g["n"]), str(g["respacing"])
    b = _tt(syn.make_batch(B, num_scene_points=N, seed=int(g["batch_seed"])))
    tab = schedule.make_tables(n, rs)
    noise = torch.from_numpy(syn.make_noise_stack(tab.num_timesteps, B, seed=int(g["noise_seed"])))
    m = _model(syn.make_sensitive_state_dict(0, n, gain=float(g["gain"])), smpl_asset, faithful=False)
    _check_out(sampler.val_losses(m, b, tab, noise, rs), g, atol=1e-4)


def test_sensitive_weights_gain_profile(sensitive_weights, synth_weights, smpl_asset):
    """What the sensitive weights are FOR: d x0 / d x_t (directional, fp64 oracle) follows the MMSE gain of a Gaussian prior - a few
    percent at t ~ n, >= 0.8 for t <= 0.1 n - while the plain random network ignores x_t at every t (~0.05).  With gain -> 1 the
    posterior mean no longer contracts an early step's rounding error (c1 J + c2 -> 1 / sqrt(alpha_t))."""
    mean, std = syn.make_body_rep_stats(0)
    B = 2
    bnp = syn.make_batch(B, 128, seed=3)
    g = np.random.Generator(np.random.PCG64(5))
    x = torch.from_numpy(g.normal(size=(B, 144)))
    d = torch.from_numpy(g.normal(size=(B, 144)))
    d = d / d.norm(dim=1, keepdim=True) * 1e-3

    def gain(sd, t):
        m = om.EgoHMROracle(sd, smpl_asset, mean, std, faithful=False, dtype=torch.float64)
        b = _tt(bnp)
        b["x_t"] = x
        a = m(b, torch.full((B,), t))["pred_x_start"]
        b["x_t"] = x + d
        return float(((m(b, torch.full((B,), t))["pred_x_start"] - a).norm(dim=1) / d.norm(dim=1)).mean())

    assert gain(synth_weights, 5) < 0.1 and gain(synth_weights, 95) < 0.1
    lo, mid, hi = gain(sensitive_weights, 5), gain(sensitive_weights, 50), gain(sensitive_weights, 99)
    assert lo >= 0.8 and hi <= 0.15 and hi < mid < lo, (lo, mid, hi)
    assert abs(mid - float(syn.mmse_gain(50 / 99))) < 0.15
