"""GPU (MI355X): the fused step launch (csrc/step.hip: output-conv responses + per-body update + the NEXT step's input conv in one launch,
one block per body; the steps' poses computed per skinning flush) against the per-step launches it replaces (gcn_out_dot_kernel,
step_body_kernel, gcn_input_kernel; models/egohmr/egohmr.py:232-260 around diffusion/gaussian_diffusion.py:298-337 / :511-556).

Same device functions, same operation order per output: every result of a sampling call must be BIT-equal between the two routes
(`EgoHMR.per_step_launches` / ehm_sample_desc.per_step_launches selects the per-step launches), on every kind of loop the sampler runs."""
import pytest
import torch

from egohmr_amd import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _model(dev, smpl_asset, **kw):
    from egohmr_amd.factory import build_synthetic_model
    m = build_synthetic_model(dev, 0, smpl_asset=smpl_asset, sensitive=dict(num_diffusion_timesteps=100), **kw)
    m.f16x3_last_steps = None
    return m


def _run(model, d, batch, noise, fused, **kw):
    model.per_step_launches = not fused                   # ehm_sample_desc.per_step_launches
    try:
        fs = model.fused_sampler
        fs.invalidate()
        r = fs.run(d, dict(batch), noise, trace=True, **kw)
        torch.cuda.synchronize()
    finally:
        model.per_step_launches = False
    o = r["other_outputs"]
    return {"sample": r["sample"].clone(), "x0": r["pred_xstart"].clone(), "verts": o["pred_vertices"].clone(), "joints": o["pred_keypoints_3d"].clone(),
            "R": o["pred_smpl_params"]["body_pose"].clone(), "orient": o["pred_smpl_params"]["global_orient"].clone(), "pose6d": o["pred_pose_6d"].clone(),
            "trace": fs.last_trace.clone()}


def _same(a, b):
    for k in a:
        assert torch.equal(a[k], b[k]), (k, float((a[k] - b[k]).abs().max()))


@pytest.mark.parametrize("B,n,respacing,ddim,precision,lowprec,all_visible", [
    (256, 100, "ddim5", True, "f16x3", None, False),     # the benchmark shape
    (40, 10, "", False, "f16x3", 3, False),               # ancestral sampling; f16 steps, the format transition (per-step launches), split-f16 steps
    (33, 50, "ddim5", True, "f16", None, False),          # plain-f16 rows, a ragged last 32-body skinning tile
    (48, 50, "ddim5", True, "f16x3", None, True),         # pass pruning: every second item all-visible (no second pass for it)
    (24, 50, "ddim5", True, "f32", None, False),          # float32 rows (the f32-input MFMA convs)
])
def test_fused_step_launch_is_bit_equal_to_the_per_step_launches(dev, smpl_asset, B, n, respacing, ddim, precision, lowprec, all_visible):
    from egohmr_amd.diffusion import create_gaussian_diffusion
    from egohmr_amd.factory import batch_to_device
    model = _model(dev, smpl_asset, diffuse_fuse=True)
    d = create_gaussian_diffusion(num_diffusion_timesteps=n, timestep_respacing=respacing)
    T = d.num_timesteps
    b = syn.make_batch(B, 512, seed=21)
    if all_visible:
        b["orig_keypoints_2d"][::2, :, 2] = 1.0
    batch = batch_to_device(b, dev)
    noise = torch.from_numpy(syn.make_noise_stack(T, B, seed=21)).to(dev)
    model.gcn_precision = precision
    model.f16x3_last_steps = (T - lowprec) if lowprec else None
    ref = _run(model, d, batch, noise, False, ddim=ddim)
    out = _run(model, d, batch, noise, True, ddim=ddim)
    assert torch.isfinite(out["verts"]).all()
    _same(out, ref)


def test_fused_step_launch_without_per_step_skinning_and_unfused_passes(dev, smpl_asset):
    """passes = 1 and lbs_every_step off (the last step takes the per-step launches: pose + skinning follow at once), also below the
    matrix-core skinning's batch size."""
    from egohmr_amd.diffusion import create_gaussian_diffusion
    from egohmr_amd.factory import batch_to_device
    model = _model(dev, smpl_asset, diffuse_fuse=False)
    model.lbs_every_step = False
    d = create_gaussian_diffusion(num_diffusion_timesteps=50, timestep_respacing="ddim5")
    for B in (5, 64):
        batch = batch_to_device(syn.make_batch(B, 256, seed=7), dev)
        noise = torch.from_numpy(syn.make_noise_stack(d.num_timesteps, B, seed=7)).to(dev)
        _same(_run(model, d, batch, noise, True, ddim=True), _run(model, d, batch, noise, False, ddim=True))


def test_fused_step_launch_under_collision_guidance(dev, smpl_asset):
    """guided steps: the gradient of step t enters the fused launch's update (gaussian_diffusion.py:378-385).  The collision gradient itself is
    scattered with float atomics (csrc/guidance.hip: the order of the additions into a vertex varies from run to run), so two runs of the SAME
    route already differ in the last bits now and then: equal within that noise here (as tests/test_gpu_api.py does for run_samples), and the
    two routes must agree no worse than a route agrees with itself."""
    from egohmr_amd.diffusion import create_gaussian_diffusion
    from egohmr_amd.factory import batch_to_device
    model = _model(dev, smpl_asset, diffuse_fuse=True)
    d = create_gaussian_diffusion(num_diffusion_timesteps=20, timestep_respacing="")
    B = 32
    batch = batch_to_device(syn.make_batch(B, 1024, seed=9), dev)
    noise = torch.from_numpy(syn.make_noise_stack(d.num_timesteps, B, seed=9)).to(dev)
    kw = dict(ddim=False, guided=True, cond_grad_weight=0.3)
    a, b, a2 = _run(model, d, batch, noise, True, **kw), _run(model, d, batch, noise, False, **kw), _run(model, d, batch, noise, True, **kw)
    assert float((a["sample"] - _run(model, d, batch, noise, True, ddim=False)["sample"]).abs().max()) > 1e-4      # the guidance is live
    for k in a:
        assert float((a[k] - b[k]).abs().max()) <= 2e-5, (k, float((a[k] - b[k]).abs().max()))
        assert float((a[k] - a2[k]).abs().max()) <= 2e-5, k
