"""GPU (MI355X) parity of the collision-guidance path: proxy loss + its vertex gradient, the LBS / rot6d
vector-Jacobian products (against torch autograd through the CPU oracle), EgoHMR.guide_coll, and the guided
DDPM loop against the reference's own run (golden g9_e2e_ddpm50_guided: reference p_sample_with_grad +
guide_coll plumbing with the same proxy plugged in as coap.collision_loss)."""
import os

import numpy as np
import pytest
import torch

from egohmr_amd import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def model(dev, synth_weights, smpl_asset):
    from egohmr_amd.factory import build_synthetic_model
    return build_synthetic_model(dev, 0, diffuse_fuse=True, state_dict=synth_weights, smpl_asset=smpl_asset)


def _posed_bodies(smpl_asset, B, seed):
    from oracle import geometry as ogeo
    from oracle.smpl import SMPLOracle
    g = np.random.Generator(np.random.PCG64(seed))
    x = torch.from_numpy(g.normal(size=(B, 144)).astype(np.float32))
    betas = torch.from_numpy(g.normal(size=(B, 10)).astype(np.float32))
    return x, betas, SMPLOracle(smpl_asset)


def test_rot6d_backward_vs_autograd(dev):
    from egohmr_amd.geometry import rot6d_to_rotmat
    from oracle import geometry as ogeo
    g = np.random.Generator(np.random.PCG64(3))
    x = torch.from_numpy(g.normal(size=(513, 6)).astype(np.float32))
    w = torch.from_numpy(g.normal(size=(513, 3, 3)).astype(np.float32))
    for mode in ("diffusion", "prohmr"):
        xc = x.clone().double().requires_grad_()
        (ogeo.rot6d_to_rotmat(xc, mode) * w.double()).sum().backward()
        xg = x.clone().to(dev).requires_grad_()
        (rot6d_to_rotmat(xg, mode) * w.to(dev)).sum().backward()
        np.testing.assert_allclose(xg.grad.cpu().numpy(), xc.grad.float().numpy(), atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize("B,N", [(3, 1500), (9, 4096)])
def test_collision_proxy_vs_oracle(dev, model, smpl_asset, B, N):
    from oracle import geometry as ogeo
    from oracle.collision import proxy_collision_loss
    x, betas, smpl = _posed_bodies(smpl_asset, B, 11)
    R = ogeo.rot6d_to_rotmat(x, "diffusion").view(B, 24, 3, 3)
    verts = smpl(betas=betas, body_pose=R[:, 1:], global_orient=R[:, [0]]).vertices
    g = np.random.Generator(np.random.PCG64(12))
    # scene: random cloud around the body plus points hugging the surface so the hinge is active
    near = verts[:, g.integers(0, 6890, size=N // 4)] + torch.from_numpy(g.normal(scale=0.02, size=(B, N // 4, 3)).astype(np.float32))
    scene = torch.cat([near, torch.from_numpy(g.uniform(-1.2, 1.2, size=(B, N - N // 4, 3)).astype(np.float32))], dim=1)
    scene[-1] = 5.0                                          # last body: nothing inside its bbox -> zero loss / gradient
    loss, gverts, _ = model.fused_sampler.collision(verts.to(dev), scene.to(dev))
    ref_loss, ref_g = [], []
    for i in range(B):
        v = verts[[i]].clone().requires_grad_()
        inds = ((scene[[i]] >= v.min(1).values.reshape(1, 3)) & (scene[[i]] <= v.max(1).values.reshape(1, 3))).all(-1)
        if inds.any():
            l = proxy_collision_loss(scene[[i]][inds].unsqueeze(0), v)
            l.backward()
            ref_loss.append(l.detach())
            ref_g.append(v.grad[0])
        else:
            ref_loss.append(torch.zeros(()))
            ref_g.append(torch.zeros(6890, 3))
    ref_loss, ref_g = torch.stack(ref_loss), torch.stack(ref_g)
    assert ref_loss[:-1].min() > 0 and ref_loss[-1] == 0
    np.testing.assert_allclose(loss.cpu().numpy(), ref_loss.numpy(), rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(gverts.cpu().numpy(), ref_g.numpy(), rtol=1e-3, atol=2e-5)


@pytest.mark.parametrize("B", [2, 11])
def test_smpl_backward_vs_autograd(dev, model, smpl_asset, B):
    """d(sum(gverts * verts))/d(pose6d) through rot6d -> chain -> pose blend -> skinning."""
    from egohmr_amd import _lib
    from oracle import geometry as ogeo
    x, betas, smpl = _posed_bodies(smpl_asset, B, 21)
    mean, std = syn.make_body_rep_stats(0)
    mean, std = torch.from_numpy(mean), torch.from_numpy(std)
    g = np.random.Generator(np.random.PCG64(22))
    gv = torch.zeros(B, 6890, 3)
    hot = g.integers(0, 6890, size=300)
    gv[:, hot] = torch.from_numpy(g.normal(size=(B, 300, 3)).astype(np.float32))
    p6 = (x * std + mean).double().requires_grad_()
    smpl64 = type(smpl)(smpl_asset, torch.float64)
    R = ogeo.rot6d_to_rotmat(p6, "diffusion").view(B, 24, 3, 3)
    (smpl64(betas=betas.double(), body_pose=R[:, 1:], global_orient=R[:, [0]]).vertices * gv.double()).sum().backward()
    out = torch.empty(B, 144, device=dev)
    L = _lib.lib()
    d_betas, d_x, d_mean, d_std, d_gv = (t.to(dev).contiguous() for t in (betas, x, mean, std, gv))   # keep alive across the call
    _lib.check(L.ehm_smpl_backward_rot6d(model.smpl.handle(), d_betas.data_ptr(), d_x.data_ptr(), d_mean.data_ptr(),
                                         d_std.data_ptr(), d_gv.data_ptr(), out.data_ptr(), B, None))
    torch.cuda.synchronize()
    ref = p6.grad.float().numpy()
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=2e-4 * np.abs(ref).max(), rtol=2e-3)


def test_guide_coll_vs_oracle(dev, model, synth_weights, smpl_asset):
    from egohmr_amd.factory import batch_to_device
    from oracle import model as om
    from oracle.collision import proxy_collision_loss
    B, N = 4, 2048
    bnp = syn.make_batch(B, N, seed=61)
    bnp["scene_pcd_verts_full"][:, : N // 3, 1] = bnp["smpl_params"]["transl"][:, None, 1] - 0.6
    mean, std = syn.make_body_rep_stats(0)
    ref = om.EgoHMROracle(synth_weights, smpl_asset, mean, std, faithful=False, collision_loss=proxy_collision_loss)
    tb = {k: ({kk: torch.from_numpy(vv) for kk, vv in v.items()} if isinstance(v, dict) else torch.from_numpy(v)) for k, v in bnp.items()}
    x_t = torch.from_numpy(syn.make_noise_stack(0, B, seed=61)[0]) * 0.5
    tb["x_t"] = x_t
    t = torch.full((B,), 3, dtype=torch.long)
    ro = ref(tb, t)
    g_ref, loss_ref = ref.guide_coll(tb, ro, t)
    assert float(g_ref.abs().max()) > 0
    gb = batch_to_device(bnp, dev)
    gb["x_t"] = x_t.to(dev)
    go = model(gb, t.to(dev))
    g = model.guide_coll(gb, go, t.to(dev), compute_grad="x_t")
    assert g.shape == (B, 144)
    scale = float(g_ref.abs().max())
    np.testing.assert_allclose(g.cpu().numpy(), g_ref.numpy(), atol=2e-3 * scale, rtol=5e-3)
    zero_joints = [0, 3, 6, 9] + list(range(12, 24))
    assert float(g.reshape(B, 24, 6)[:, zero_joints].abs().max()) == 0.0


@pytest.mark.parametrize("route", ["fused", "generic"])
def test_guided_ddpm_vs_reference_golden(golden_dir, dev, model, route):
    from egohmr_amd.diffusion import create_gaussian_diffusion
    from egohmr_amd.factory import batch_to_device
    g = np.load(os.path.join(golden_dir, "g9_e2e_ddpm50_guided.npz"))
    B, N, n = int(g["B"]), int(g["N"]), int(g["n"])
    d = create_gaussian_diffusion(num_diffusion_timesteps=n, timestep_respacing="")
    bnp = syn.make_batch(B, num_scene_points=N, seed=int(g["batch_seed"]))
    bnp["scene_pcd_verts_full"][:, : N // 3, 1] = bnp["smpl_params"]["transl"][:, None, 1] - 0.6
    b = batch_to_device(bnp, dev)
    noise = torch.from_numpy(syn.make_noise_stack(d.num_timesteps, B, seed=int(g["noise_seed"]))).to(dev)
    w = float(g["cond_grad_weight"])
    if route == "fused":
        res = model.fused_sampler.run(d, b, noise, ddim=False, guided=True, cond_grad_weight=w, trace=True)
        o = res["other_outputs"]
        tr = model.fused_sampler.last_trace.cpu().numpy()
        low = model.fused_sampler.last_lowprec                                        # leading steps on plain f16 operands (calibrated precision schedule)
        assert d.num_timesteps - low >= 11 + 8                                        # every guided step and at least the 8 steps before them run in f16x3
        np.testing.assert_allclose(tr[:low + 1], g["x_t_trace"][:low + 1], atol=2e-3)
        np.testing.assert_allclose(tr[low + 4:], g["x_t_trace"][low + 4:], atol=3e-4)
        np.testing.assert_allclose(tr[-1], g["x_t_trace"][-1], atol=5e-5)
    else:
        d.allow_fused = False
        o = d.val_losses(model, b, shape=[B, 144], clip_denoised=False, timestep_respacing="", compute_loss=False,
                         cond_fn_with_grad=True, cond_grad_weight=w, noise_stack=noise)
    c = lambda t: t.detach().cpu().numpy()
    np.testing.assert_allclose(c(o["pred_x_start"]), g["pred_x_start"], atol=2e-4)
    np.testing.assert_allclose(c(o["pred_vertices"][:, :64]), g["verts_head"], atol=1e-4)
    np.testing.assert_allclose(c(o["pred_keypoints_3d"]), g["joints"], atol=1e-4)


def test_guided_ddim_vs_reference_golden(golden_dir, dev, model):
    """ddim_sample_with_grad (gaussian_diffusion.py:559-614: collision gradient through eps on the last four respaced steps)
    against the reference's own run, golden g12; and `val_losses(..., 'ddim10', cond_fn_with_grad=True)` routes to it."""
    from egohmr_amd.diffusion import create_gaussian_diffusion
    from egohmr_amd.factory import batch_to_device
    g = np.load(os.path.join(golden_dir, "g12_e2e_ddim10_guided.npz"))
    B, N, n, rs = int(g["B"]), int(g["N"]), int(g["n"]), str(g["respacing"])
    d = create_gaussian_diffusion(num_diffusion_timesteps=n, timestep_respacing=rs)
    bnp = syn.make_batch(B, num_scene_points=N, seed=int(g["batch_seed"]))
    bnp["scene_pcd_verts_full"][:, : N // 3, 1] = bnp["smpl_params"]["transl"][:, None, 1] - 0.6
    b = batch_to_device(bnp, dev)
    noise = torch.from_numpy(syn.make_noise_stack(d.num_timesteps, B, seed=int(g["noise_seed"]))).to(dev)
    xs = [noise[0].cpu().numpy()]
    for out in d.ddim_sample_loop_progressive(model, b, [B, 144], cond_fn_with_grad=True, noise_stack=noise):
        xs.append(out["sample"].cpu().numpy())
    np.testing.assert_allclose(np.stack(xs[:-1]), g["x_t_trace"], atol=2e-4)
    c = lambda t: t.detach().cpu().numpy()
    # both routes of the loop: the one-call native loop (round 6: the guided DDIM update lives in step_body_one) and the python-driven generic one
    for fused in (True, False):
        d.allow_fused = fused
        o = d.val_losses(model, b, shape=[B, 144], clip_denoised=False, timestep_respacing=rs, compute_loss=False,
                         cond_fn_with_grad=True, noise_stack=noise)
        np.testing.assert_allclose(c(o["pred_x_start"]), g["pred_x_start"], atol=2e-4)
        np.testing.assert_allclose(c(o["pred_vertices"][:, :64]), g["verts_head"], atol=1e-4)
        np.testing.assert_allclose(c(o["pred_keypoints_3d"]), g["joints"], atol=1e-4)
    d.allow_fused = True
    fs = model.fused_sampler
    r = fs.run(d, dict(b), noise, ddim=True, guided=True, cond_grad_weight=1.0, trace=True)
    np.testing.assert_allclose(fs.last_trace.cpu().numpy(), g["x_t_trace"], atol=2e-4)          # x_t fed to every step, guided tail included
    np.testing.assert_allclose(c(r["sample"]), xs[-1], atol=2e-5)                               # the final sample of the generic route
    assert torch.equal(r["pred_xstart"], r["sample"])                                           # the guided x0 of the last step (alpha_bar_prev = 1)
    ru = fs.run(d, dict(b), noise, ddim=True, guided=False)
    assert float((ru["sample"] - r["sample"]).abs().max()) > 2e-5                               # the guidance is live in the one-call loop
    # the guidance was live on the last steps only
    xs0 = [out["sample"].cpu().numpy() for out in d.ddim_sample_loop_progressive(model, b, [B, 144], cond_fn_with_grad=False, noise_stack=noise)]
    assert np.abs(xs0[-1] - xs[-1]).max() > 2e-5 and np.abs(xs0[5] - xs[6]).max() == 0.0
