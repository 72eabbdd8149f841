"""GPU (MI355X): edge cases of the sampling path - a single item with the reference's real scene size (N = 20000,
preprocess_scene_s2_for_test.py:24), and collision guidance with nothing to collide with (the reference's zero-loss short-circuit,
egohmr.py:561,569-570: gradient = zeros, so the guided loop must reproduce the unguided one)."""
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


def test_single_item_with_20000_scene_points_vs_oracle(dev, model, synth_weights, smpl_asset):
    from egohmr_amd.diffusion import create_gaussian_diffusion
    from egohmr_amd.factory import batch_to_device
    from oracle import model as om, sampler as osamp, schedule as osched
    B, N, n, rs = 1, 20000, 50, "ddim5"
    bnp = syn.make_batch(B, N, seed=81)
    d = create_gaussian_diffusion(num_diffusion_timesteps=n, timestep_respacing=rs)
    noise = syn.make_noise_stack(d.num_timesteps, B, seed=81)
    out = d.val_losses(model, batch_to_device(bnp, dev), shape=[B, 144], clip_denoised=False, timestep_respacing=rs, compute_loss=False,
                       noise_stack=torch.from_numpy(noise).to(dev))
    mean, std = syn.make_body_rep_stats(0)
    ref = om.EgoHMROracle(synth_weights, smpl_asset, mean, std, faithful=False)
    tb = {k: ({kk: torch.from_numpy(vv) for kk, vv in v.items()} if isinstance(v, dict) else torch.from_numpy(v)) for k, v in bnp.items()}
    ro = osamp.val_losses(ref, tb, osched.make_tables(n, rs), torch.from_numpy(noise), rs)
    assert out["pred_vertices"].shape == (1, 6890, 3) and out["pred_keypoints_3d"].shape == (1, 45, 3)
    np.testing.assert_allclose(out["pred_vertices"].cpu().numpy(), ro["pred_vertices"].numpy(), atol=1e-4)
    np.testing.assert_allclose(out["pred_keypoints_3d"].cpu().numpy(), ro["pred_keypoints_3d"].numpy(), atol=1e-4)
    np.testing.assert_allclose(out["pred_smpl_params"]["betas"].cpu().numpy(), ro["pred_smpl_params"]["betas"].numpy(), atol=5e-5)


@pytest.mark.parametrize("volsmpl", [False, True])
def test_guidance_with_nothing_to_collide_with_equals_the_unguided_loop(dev, synth_weights, smpl_asset, volsmpl):
    """Scene 50 m away: no scene point inside any body's bounding box / within tau of it -> every item's loss is 0 -> the reference returns a
    zero gradient (egohmr.py:561,569-570; egohmr_volsmpl.py:617-629) and the mean shift of p_sample_with_grad vanishes: bit-equal bodies."""
    from egohmr_amd.diffusion import create_gaussian_diffusion
    from egohmr_amd.factory import batch_to_device, build_synthetic_model
    m = build_synthetic_model(dev, 0, diffuse_fuse=True, state_dict=synth_weights, smpl_asset=smpl_asset, volsmpl=volsmpl)
    m.f16x3_last_steps = None                        # same arithmetic on every step in both loops (a guided loop calibrates its own schedule)
    B, N = 5, 2048
    bnp = syn.make_batch(B, N, seed=82)
    bnp["scene_pcd_verts_full"] = bnp["scene_pcd_verts_full"] + np.float32(50.0)
    b = batch_to_device(bnp, dev)
    d = create_gaussian_diffusion(num_diffusion_timesteps=50, timestep_respacing="")
    noise = torch.from_numpy(syn.make_noise_stack(d.num_timesteps, B, seed=82)).to(dev)
    w = 30.0 if volsmpl else 2.0
    guided = d.p_sample_loop(m, dict(b), [B, 144], cond_fn_with_grad=True, cond_grad_weight=w, noise_stack=noise)
    plain = d.p_sample_loop(m, dict(b), [B, 144], cond_fn_with_grad=False, noise_stack=noise)
    assert torch.equal(guided["sample"], plain["sample"])
    assert torch.equal(guided["other_outputs"]["pred_vertices"], plain["other_outputs"]["pred_vertices"])
    assert m.eval_coll(guided["other_outputs"]) == [0.0] * B
    g = m.guide_coll(b | {"x_t": noise[0]}, guided["other_outputs"], torch.zeros(B, dtype=torch.long, device=dev))
    assert g.shape == (B, 144) and float(g.abs().max()) == 0.0


def test_resnet_activation_offsets_beyond_2_gib(dev, synth_weights):
    """The X2 trunk addresses its input activations through 32-bit BYTE offsets in a buffer descriptor (csrc/conv.hip): a batch whose
    layer-1 tensor is larger than 2 GiB (1100 images: 883 M elements = 3.5 GB) must give the same features as the same images in a
    small batch, and a batch past the 2^30-element limit must be refused, not silently mis-addressed."""
    from egohmr_amd import _lib
    from egohmr_amd.encoders import ResNet50Features
    net = ResNet50Features()
    net.load_state_dict({k[len("backbone."):]: torch.from_numpy(np.asarray(v)) for k, v in synth_weights.items() if k.startswith("backbone.")})
    net = net.to(dev).eval()
    fwd = net.folded()
    g = torch.Generator(device=dev).manual_seed(5)
    img = torch.randn(1100, 3, 224, 224, device=dev, generator=g)
    with torch.no_grad():
        big = fwd(img)
        small = fwd(img[-4:].contiguous())
        first = fwd(img[:4].contiguous())
    np.testing.assert_allclose(big[-4:].cpu().numpy(), small.cpu().numpy(), atol=5e-6)
    np.testing.assert_allclose(big[:4].cpu().numpy(), first.cpu().numpy(), atol=5e-6)
    assert torch.isfinite(big).all() and float(big.abs().max()) > 0
    del big, img
    torch.cuda.empty_cache()
    with pytest.raises(_lib.EgoHMRHipError):
        fwd(torch.zeros(1400, 3, 224, 224, device=dev))


def test_timestep_vectors_against_the_reference_embedding(dev, model, golden_dir):
    """Product side of G5 (models/egohmr/egohmr.py:642-643 TimestepEmbedder, generated by importing the reference): the module's
    embedding itself, and FusedSampler.timestep_vectors = that embedding through the timestep slice of the folded input conv -
    for a tensor of timesteps, for a python list (the cached route the sampling loop takes) and for the cache hit."""
    import os
    g = np.load(os.path.join(golden_dir, "g5_timestep_embed.npz"))
    t = torch.from_numpy(g["t"]).to(dev)
    with torch.no_grad():
        emb = model.embed_timestep.time_embed(model.sequence_pos_encoder.pe[t][:, 0])
    np.testing.assert_allclose(emb.cpu().numpy(), g["emb"].reshape(len(g["t"]), -1), atol=1e-6)
    fs = model.fused_sampler
    fs.gcn()
    want = (torch.einsum("ne,kef->nkf", torch.from_numpy(g["emb"].reshape(len(g["t"]), -1)).to(dev).double(), fs._folded.W_t) + fs._folded.bx[None]).float()
    tv = fs.timestep_vectors(t)
    tl = fs.timestep_vectors([int(v) for v in g["t"]])
    again = fs.timestep_vectors([int(v) for v in g["t"]])
    scale = float(want.abs().max())
    assert float((tv - want).abs().max()) <= 2e-6 * max(scale, 1.0)
    assert torch.equal(tl, tv) and again is tl                       # the list route is the same arithmetic, then a cache hit
    other = fs.timestep_vectors([0, 1])
    assert other.shape[0] == 2 and torch.equal(other, tv[:2])        # a different sequence replaces the cache entry


def test_module_calls_are_the_hip_path(dev, model, golden_dir):
    """model.backbone(img) and model.scene_enc(pts) run the HIP kernels (G6 from the reference); the parameter containers below them refuse
    to run eager arithmetic, and CPU tensors are refused - there is no second implementation to fall into."""
    import os
    from egohmr_amd import _lib
    g = np.load(os.path.join(golden_dir, "g6_resnet50.npz"))
    rng = np.random.Generator(np.random.PCG64(int(g["img_seed"])))
    rng.uniform(-1, 1, size=(2, 257, 3))
    img = torch.from_numpy(rng.normal(size=(2, 3, 224, 224)).astype(np.float32)).to(dev)
    with torch.no_grad():
        out = model.backbone(img)
    np.testing.assert_allclose(out.cpu().numpy(), g["feat"], atol=3e-5)
    assert model.backbone.current() is model.fused_sampler._backbone_fn()          # one packed copy of the weights
    with pytest.raises(_lib.EgoHMRHipError):
        model.backbone(img.cpu())
    with pytest.raises(_lib.EgoHMRHipError):
        model.backbone.layer1[0](torch.zeros(1, 64, 56, 56, device=dev))
    with pytest.raises(_lib.EgoHMRHipError):
        model.scene_enc.block_0(torch.zeros(1, 8, 256, device=dev))
    p = np.load(os.path.join(golden_dir, "g6_pointnet.npz"))
    with torch.no_grad():
        c = model.scene_enc(torch.from_numpy(p["pts"]).to(dev))
    np.testing.assert_allclose(c.cpu().numpy(), p["feat"], atol=2e-5)


@pytest.mark.parametrize("B,lbs_every_step", [(40, True), (40, False), (8, True)])
def test_sampler_with_a_dense_skinning_weights_body_model(dev, synth_weights, smpl_asset, B, lbs_every_step):
    """A body model whose vertices carry more than four skinning weights has no matrix-core skinning fragments (with B >= 24 and
    lbs_every_step the loop used to pick deferred MFMA skinning from B alone and fail with EINVAL).  The sampling loop must fall back to the
    VALU skinning inside every step and still match the oracle (which uses the same dense weights)."""
    from egohmr_amd.diffusion import create_gaussian_diffusion
    from egohmr_amd.factory import batch_to_device, build_synthetic_model
    from oracle import model as om, sampler as osamp, schedule as osched
    asset = dict(smpl_asset)
    g = np.random.Generator(np.random.PCG64(77))
    w = np.array(asset["lbs_weights"], dtype=np.float64).copy()
    for v in g.choice(w.shape[0], size=w.shape[0] // 3, replace=False):
        js = g.choice(w.shape[1], size=6, replace=False)
        w[v] = 0.0
        w[v, js] = g.uniform(0.05, 1.0, size=6)
        w[v] /= w[v].sum()
    asset["lbs_weights"] = w.astype(np.float32)
    m = build_synthetic_model(dev, 0, diffuse_fuse=True, state_dict=synth_weights, smpl_asset=asset)
    m.lbs_every_step = lbs_every_step
    N, n, rs = 512, 50, "ddim5"
    bnp = syn.make_batch(B, N, seed=83)
    d = create_gaussian_diffusion(num_diffusion_timesteps=n, timestep_respacing=rs)
    noise = syn.make_noise_stack(d.num_timesteps, B, seed=83)
    out = d.val_losses(m, batch_to_device(bnp, dev), shape=[B, 144], clip_denoised=False, timestep_respacing=rs, compute_loss=False,
                       noise_stack=torch.from_numpy(noise).to(dev))
    mean, std = syn.make_body_rep_stats(0)
    ref = om.EgoHMROracle(synth_weights, asset, mean, std, faithful=False)
    tb = {k: ({kk: torch.from_numpy(vv) for kk, vv in v.items()} if isinstance(v, dict) else torch.from_numpy(v)) for k, v in bnp.items()}
    ro = osamp.val_losses(ref, tb, osched.make_tables(n, rs), torch.from_numpy(noise), rs)
    np.testing.assert_allclose(out["pred_vertices"].cpu().numpy(), ro["pred_vertices"].numpy(), atol=1e-4)
    np.testing.assert_allclose(out["pred_keypoints_3d"].cpu().numpy(), ro["pred_keypoints_3d"].numpy(), atol=1e-4)
