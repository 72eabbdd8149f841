"""GPU (MI355X): a NaN / Inf in ONE item's inputs.  The reference's float32 graph carries it into every output of that item (ReLU, max-pool and
0 * NaN propagate in torch: resnet.py / resnet_pointnet.py / gaussian_diffusion.py:357-359) and leaves the other items alone; the kernels'
v_max and saturating conversions would swallow it, so the flag travels beside the data (FusedSampler.prepare, EgoHMR._pack_output)."""
import numpy as np
import pytest
import torch

from egohmr_amd import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def model(dev, synth_weights, smpl_asset):
    from egohmr_amd.factory import build_synthetic_model
    return build_synthetic_model(dev, 0, diffuse_fuse=True, state_dict=synth_weights, smpl_asset=smpl_asset)


def _poison(kind):
    def f(b, noise):
        if kind == "image_nan":
            b["img"][1, 2, 100, 7] = np.nan
        elif kind == "scene_inf":
            b["scene_pcd_verts_full"][1, 33, 0] = np.inf
        elif kind == "fx_nan":
            b["fx"][1] = np.nan
        elif kind == "x_T_nan":
            noise[0, 1, 140] = np.nan
        elif kind == "mid_step_noise_inf":
            noise[4, 1, 0] = np.inf
    return f


@pytest.mark.parametrize("kind", ["image_nan", "scene_inf", "fx_nan", "x_T_nan", "mid_step_noise_inf"])
@pytest.mark.parametrize("route", ["fused", "stepwise"])
def test_non_finite_input_of_one_item_becomes_nan_outputs_of_that_item_only(dev, model, kind, route):
    from egohmr_amd.diffusion import create_gaussian_diffusion
    from egohmr_amd.factory import batch_to_device
    B, N, n = 3, 512, 10
    d = create_gaussian_diffusion(num_diffusion_timesteps=n, timestep_respacing="")

    def run(poison):
        bnp = syn.make_batch(B, N, seed=91)
        noise = syn.make_noise_stack(d.num_timesteps, B, seed=91)
        if poison:
            _poison(kind)(bnp, noise)
        d.allow_fused = route == "fused"
        return d.p_sample_loop(model, batch_to_device(bnp, dev), [B, 144], noise_stack=torch.from_numpy(noise).to(dev))

    clean, dirty = run(False), run(True)
    oc, od = clean["other_outputs"], dirty["other_outputs"]
    pairs = [(clean["sample"], dirty["sample"]), (oc["pred_vertices"], od["pred_vertices"]), (oc["pred_keypoints_3d"], od["pred_keypoints_3d"]),
             (oc["pred_keypoints_2d_full"], od["pred_keypoints_2d_full"]), (oc["pred_smpl_params"]["body_pose"], od["pred_smpl_params"]["body_pose"])]
    if kind in ("image_nan", "scene_inf", "fx_nan"):                      # betas come from the conditioning only
        pairs.append((oc["pred_smpl_params"]["betas"], od["pred_smpl_params"]["betas"]))
    for c, x in pairs:
        assert torch.isfinite(c).all()
        assert torch.isnan(x[1]).all(), "the poisoned item must come out as NaN"
        assert torch.equal(x[[0, 2]], c[[0, 2]]), "the other items must not change"


@pytest.mark.parametrize("route", ["fused", "stepwise"])
def test_nan_in_the_last_steps_noise_lands_in_that_element_of_the_sample_only(dev, model, route):
    """gaussian_diffusion.py:357-359: sample = mean + nonzero_mask * exp(0.5 * log_variance) * noise with nonzero_mask = 0 at t = 0: 0 * NaN = NaN
    in that element; the model outputs of the last step were computed before the draw."""
    from egohmr_amd.diffusion import create_gaussian_diffusion
    from egohmr_amd.factory import batch_to_device
    B, N, n = 3, 512, 10
    d = create_gaussian_diffusion(num_diffusion_timesteps=n, timestep_respacing="")
    d.allow_fused = route == "fused"

    def run(poison):
        noise = syn.make_noise_stack(d.num_timesteps, B, seed=92)
        if poison:
            noise[-1, 1, 5] = np.nan
        return d.p_sample_loop(model, batch_to_device(syn.make_batch(B, N, seed=92), dev), [B, 144], noise_stack=torch.from_numpy(noise).to(dev))

    clean, dirty = run(False), run(True)
    hole = torch.zeros(B, 144, dtype=torch.bool, device=dev)
    hole[1, 5] = True
    assert torch.equal(torch.isnan(dirty["sample"]), hole)
    assert torch.equal(dirty["sample"][~hole], clean["sample"][~hole])
    assert torch.equal(dirty["other_outputs"]["pred_vertices"], clean["other_outputs"]["pred_vertices"])
    assert torch.equal(dirty["pred_xstart"], clean["pred_xstart"])
