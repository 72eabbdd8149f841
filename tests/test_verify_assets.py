"""tools/verify_assets.py - the self-check a maintainer runs on the real SMPL pkl + checkpoint - exercised on synthetic stand-ins:
the structural checks on the CPU, the whole tool (SMPL pkl through the chumpy-free unpickler, checkpoint through io.load_checkpoint,
gain / calibration / precision table) on the GPU."""
import importlib.util
import json
import os
import pickle

import numpy as np
import pytest
import torch

from egohmr_amd import synthetic as syn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("verify_assets", os.path.join(REPO, "tools", "verify_assets.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _write_smpl_pkl(path, asset):
    """The layout of the official SMPL_*.pkl as far as the loader reads it (no chumpy objects: plain arrays)."""
    V = asset["v_template"].shape[0]
    kin = np.stack([np.where(asset["parents"] < 0, 2 ** 32 - 1, asset["parents"]).astype(np.int64), np.arange(24)])
    d = {"v_template": asset["v_template"], "shapedirs": asset["shapedirs"], "posedirs": asset["posedirs"].T.reshape(V, 3, 207),
         "J_regressor": asset["J_regressor"], "weights": asset["lbs_weights"], "kintree_table": kin, "f": asset["faces"]}
    with open(path, "wb") as f:
        pickle.dump(d, f, protocol=2)


def test_structural_checks_accept_the_synthetic_asset_and_catch_a_broken_one(smpl_asset):
    va = _tool()
    rep = va.check_smpl_asset(smpl_asset)
    assert all(ok for ok, _ in rep.values()), {k: v for k, v in rep.items() if not v[0]}
    bad = dict(smpl_asset)
    bad["lbs_weights"] = smpl_asset["lbs_weights"] * 0.5                    # no longer a partition of unity
    bad["J_regressor"] = smpl_asset["J_regressor"].copy()
    bad["J_regressor"][3] *= 2.0
    rep = va.check_smpl_asset(bad)
    assert not rep["skinning_partition_of_unity"][0] and not rep["j_regressor_rows_sum_to_1"][0]


@pytest.mark.gpu
def test_verify_assets_end_to_end_on_synthetic_files(tmp_path, smpl_asset, capsys):
    va = _tool()
    pkl = tmp_path / "smpl" / "SMPL_NEUTRAL.pkl"
    os.makedirs(pkl.parent)
    _write_smpl_pkl(pkl, smpl_asset)
    sd = syn.make_sensitive_state_dict(0, 50)
    ckpt = tmp_path / "best_model.pt"
    torch.save({"state_dict": {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, "epoch": 1}, ckpt)
    mean, std = syn.make_body_rep_stats(0)
    np.savez(tmp_path / "preprocess_stats.npz", Xmean=mean, Xstd=std)
    rc = va.main([str(tmp_path), str(ckpt), "--stats", str(tmp_path / "preprocess_stats.npz"), "--timesteps", "50", "--batch", "8", "--scene-points", "512", "--json"])
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert rc == 0, out["hard_failures"]
    r = out["report"]
    assert r["checkpoint"]["no_unexpected_keys"]["ok"] and r["smpl"]["identity_pose_reproduces_v_shaped_on_gpu"]["ok"]
    assert r["precision"]["f16x3_within_1e-4_of_f32"]["ok"] and r["precision"]["calibrated_f16x3_last_steps"]["T"] == 50
    assert float(r["precision"]["measured_gain_dx0_dxt"]["0"]) > 0.8           # the sensitive weights: gain ~ 1 at t = 0
    # a checkpoint with a stray key is reported
    sd2 = dict(sd)
    sd2["diffusion_model.not_a_layer.weight"] = np.zeros(3, np.float32)
    torch.save({"state_dict": {k: torch.from_numpy(np.asarray(v)) for k, v in sd2.items()}}, ckpt)
    rc = va.main([str(tmp_path), str(ckpt), "--timesteps", "50", "--batch", "8", "--scene-points", "512", "--json"])
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert rc == 1 and any("no_unexpected_keys" in f for f in out["hard_failures"])
