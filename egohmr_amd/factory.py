"""Convenience constructors used by bench.py, __graft_entry__.py and the tests."""
from __future__ import annotations

import numpy as np
import torch

from . import synthetic as syn
from .diffusion import create_gaussian_diffusion
from .model import EgoHMR, EgoHMRVolsmpl


def build_synthetic_model(device="cuda", seed: int = 0, diffuse_fuse: bool = True, identity_stats: bool = False,
                          state_dict: dict | None = None, smpl_asset: dict | None = None, gcn_nonlocal_layer: bool = False,
                          volsmpl: bool = False, sensitive=None) -> EgoHMR:
    """EgoHMR with the test-time flags of test_egohmr.py:112-118, seeded synthetic weights and SMPL asset
    (no checkpoint / licensed model file exists offline).  sensitive = None: the plain random network (ignores x_t);
    True / dict(num_diffusion_timesteps=, gain=, prior_var=): the x_t-sensitive, trained-like denoiser of
    synthetic.make_sensitive_state_dict."""
    mean, std = syn.make_body_rep_stats(seed, identity=identity_stats)
    model = (EgoHMRVolsmpl if volsmpl else EgoHMR)(device=device, body_rep_mean=mean, body_rep_std=std, with_focal_length=True, with_bbox_info=True,
                   with_cam_center=True, scene_feat_dim=512, scene_type="cube", scene_cano=True, cond_mask_prob=0.0,
                   only_mask_img_cond=True, pelvis_vis_loosen=True, diffuse_fuse=diffuse_fuse, gcn_nonlocal_layer=gcn_nonlocal_layer,
                   smpl_asset=smpl_asset if smpl_asset is not None else syn.make_smpl_asset(seed))
    if state_dict is not None:
        sd = state_dict
    elif sensitive:
        sd = syn.make_sensitive_state_dict(seed, **(sensitive if isinstance(sensitive, dict) else {}), nonlocal_layer=gcn_nonlocal_layer)
    else:
        sd = syn.make_state_dict(seed, nonlocal_layer=gcn_nonlocal_layer)
    res = model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=False)
    assert not res.unexpected_keys and all(k.startswith("smpl") for k in res.missing_keys), res
    model.eval()
    return model


def batch_to_device(batch: dict, device) -> dict:
    """utils/other_utils.py recursive_to for the numpy batches of synthetic.make_batch."""
    out = {}
    for k, v in batch.items():
        if isinstance(v, dict):
            out[k] = batch_to_device(v, device)
        else:
            out[k] = (torch.from_numpy(np.asarray(v)) if not torch.is_tensor(v) else v).to(device)
    return out
