"""Files on either side of the sampling path.

* `save_results` / `load_results`: the `results_seed_{seed}.pkl` dictionary test_egohmr.py:672-695 writes (pickle protocol 2,
  numpy arrays, exactly these keys) - what downstream visualisation / evaluation scripts of the reference read.
* `load_stage1_cam`: `results.pkl['pred_cam_full_list']` written by the stage-1 script (test_prohmr_scene.py:417-426) and
  consumed by `--two_stage` runs.
* `load_preprocess_stats`, `load_smpl_mean_params`, `load_checkpoint`: test_egohmr.py:109-111, models/egohmr/egohmr.py:669-671,
  test_egohmr.py:125-127.

Pure host-side numpy / pickle code: nothing here is on the measured path."""
from __future__ import annotations

import os
import pickle
from typing import Mapping

import numpy as np
import torch

RESULT_KEYS = ("pred_betas_list", "pred_global_orient_list", "pred_body_pose_list", "collision_ratio_list", "contact_ratio_list",
               "gt_cam_full_list")                  # + 'pred_cam_full_list' for --two_stage runs


def _np(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def results_dict(pred_betas, pred_global_orient, pred_body_pose, collision_ratio, contact_ratio, gt_cam_full, pred_cam_full=None):
    """Assemble the dictionary of test_egohmr.py:685-693.  Shapes as the reference concatenates them:
    pred_betas [n, S, 10], pred_global_orient [n, S, 1, 3, 3], pred_body_pose [n, S, 23, 3, 3], collision / contact ratio [n, S],
    gt_cam_full [n, 3] (pred_cam_full [n, 3] when stage 1 supplied the translation)."""
    d = {"pred_betas_list": _np(pred_betas), "pred_global_orient_list": _np(pred_global_orient),
         "pred_body_pose_list": _np(pred_body_pose), "collision_ratio_list": _np(collision_ratio),
         "contact_ratio_list": _np(contact_ratio)}
    if pred_cam_full is not None:
        d["pred_cam_full_list"] = _np(pred_cam_full)
    d["gt_cam_full_list"] = _np(gt_cam_full)
    return d


def save_results(save_root: str, model_id: str, seed: int, results: Mapping[str, np.ndarray]) -> str:
    """`{save_root}/output_egohmr_{model_id}/results_seed_{seed}.pkl`, pickle protocol 2 (test_egohmr.py:680-695)."""
    missing = [k for k in RESULT_KEYS if k not in results]
    if missing:
        raise KeyError(f"results are missing {missing}")
    folder = os.path.join(save_root, f"output_egohmr_{model_id}")
    os.makedirs(folder, exist_ok=True)
    path = os.path.join(folder, f"results_seed_{seed}.pkl")
    with open(path, "wb") as f:
        pickle.dump({k: _np(v) for k, v in results.items()}, f, protocol=2)
    return path


def load_results(path: str) -> dict:
    with open(path, "rb") as f:
        d = pickle.load(f, encoding="latin1")
    missing = [k for k in RESULT_KEYS if k not in d]
    if missing:
        raise KeyError(f"{path}: not a results_seed_*.pkl (missing {missing})")
    return d


def load_stage1_cam(path: str) -> np.ndarray:
    """`pred_cam_full_list` [n, 3] of the stage-1 `results.pkl` (test_prohmr_scene.py:417-426), the body translation a
    `--two_stage` stage-2 run conditions on (test_egohmr.py:100-104)."""
    with open(path, "rb") as f:
        d = pickle.load(f, encoding="latin1")
    if "pred_cam_full_list" not in d:
        raise KeyError(f"{path}: no 'pred_cam_full_list' (is this a stage-1 results.pkl?)")
    cam = np.asarray(d["pred_cam_full_list"], dtype=np.float32)
    if cam.ndim != 2 or cam.shape[1] != 3:
        raise ValueError(f"{path}: pred_cam_full_list has shape {cam.shape}, expected [n, 3]")
    return cam


def load_preprocess_stats(path: str):
    """`preprocess_stats.npz` -> (body_rep_mean [144], body_rep_std [144]) (test_egohmr.py:109-111: keys 'Xmean', 'Xstd')."""
    z = np.load(path)
    mean, std = np.asarray(z["Xmean"], np.float32).reshape(-1), np.asarray(z["Xstd"], np.float32).reshape(-1)
    if mean.shape != (144,) or std.shape != (144,):
        raise ValueError(f"{path}: Xmean/Xstd have shapes {mean.shape}/{std.shape}, expected 144 values each")
    return mean, std


def load_smpl_mean_params(path: str) -> np.ndarray:
    """`data/smpl_mean_params.npz['shape']` -> init betas [1, 10] (models/egohmr/egohmr.py:669-671)."""
    shape = np.asarray(np.load(path)["shape"], np.float32).reshape(1, -1)
    if shape.shape != (1, 10):
        raise ValueError(f"{path}: 'shape' has {shape.size} values, expected 10")
    return shape


def load_checkpoint(model: torch.nn.Module, path_or_state, strict: bool = False):
    """`weights = torch.load(ckpt); model.load_state_dict(weights['state_dict'], strict=False)` (test_egohmr.py:125-127).
    Returns torch's (missing_keys, unexpected_keys) so a caller can see what a checkpoint did not cover."""
    w = torch.load(path_or_state, map_location="cpu") if isinstance(path_or_state, (str, os.PathLike)) else path_or_state
    sd = w["state_dict"] if isinstance(w, Mapping) and "state_dict" in w else w
    return model.load_state_dict(sd, strict=strict)
