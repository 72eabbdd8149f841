"""Evaluation metrics: CPU oracle vs the reference's golden, GPU path vs the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import metrics as om


def test_oracle_procrustes_vs_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "g11_procrustes.npz"))
    np.testing.assert_allclose(om.pa_mpjpe(g["pred"], g["gt"]), g["pa_mpjpe"], rtol=1e-10, atol=1e-12)


def test_oracle_nn_dist2_definition():
    g = np.random.Generator(np.random.PCG64(1))
    x, y = g.normal(size=(2, 50, 3)), g.normal(size=(2, 70, 3))
    d, idx = om.nn_dist2(x, y)
    for b in range(2):
        for i in range(50):
            full = ((x[b, i] - y[b]) ** 2).sum(-1)
            assert idx[b, i] == full.argmin() and abs(d[b, i] - full.min()) < 1e-12


@pytest.mark.gpu
def test_pa_mpjpe_device_vs_reference_golden(golden_dir):
    from egohmr_amd import metrics
    g = np.load(os.path.join(golden_dir, "g11_procrustes.npz"))
    dev = torch.device("cuda:0")
    out = metrics.pa_mpjpe(torch.from_numpy(g["pred"]).float().to(dev), torch.from_numpy(g["gt"]).float().to(dev))
    np.testing.assert_allclose(out.cpu().numpy(), g["pa_mpjpe"], rtol=2e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("B,P1,P2", [(1, 1, 1), (2, 6890, 4096), (3, 257, 20000), (2, 300, 2049)])
def test_nn_dist2_device_vs_oracle(B, P1, P2):
    """knn_points(K=1) replacement incl. ragged tile sizes (P2 not a multiple of the 2048-point LDS tile) and 1-point clouds."""
    from egohmr_amd import metrics
    g = np.random.Generator(np.random.PCG64(B * 1000 + P1))
    x = g.uniform(-1, 1, size=(B, P1, 3)).astype(np.float32)
    y = g.uniform(-1, 1, size=(B, P2, 3)).astype(np.float32)
    dev = torch.device("cuda:0")
    d, idx = metrics.nn_dist2(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev), return_idx=True)
    rd, ridx = om.nn_dist2(x, y)
    np.testing.assert_allclose(d.cpu().numpy(), rd, rtol=1e-5, atol=1e-7)
    picked = np.take_along_axis(((x[:, :, None, :].astype(np.float64) - y[:, None, :, :]) ** 2).sum(-1), idx.cpu().numpy()[..., None].astype(np.int64), 2)[..., 0] \
        if P1 * P2 < 5e6 else rd
    np.testing.assert_allclose(picked, rd, rtol=1e-5, atol=1e-7)          # the returned index attains the minimum
    cx, cy, cn = metrics.chamfer_distance(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev))
    assert cn is None and cx.shape == (B, P1) and cy.shape == (B, P2)
    np.testing.assert_allclose(cy.cpu().numpy(), om.nn_dist2(y, x)[0], rtol=1e-5, atol=1e-7)
    contact = metrics.contact_score(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev), 0.02)
    np.testing.assert_array_equal(contact.cpu().numpy(), rd.min(-1) < 0.02)


@pytest.mark.gpu
def test_mpjpe_v2v_formulas():
    from egohmr_amd import metrics
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    p, q = torch.randn(4, 3, 45, 3, generator=g).to(dev), torch.randn(4, 1, 45, 3, generator=g).to(dev)
    ref = torch.sqrt((((p[..., :24, :] - p[..., :1, :]) - (q[..., :24, :] - q[..., :1, :])) ** 2).sum(-1)).mean(-1)
    assert torch.allclose(metrics.mpjpe(p, q), ref)
    assert torch.allclose(metrics.mpjpe(p + 5.0, q), ref, atol=1e-5)       # translation invariant (pelvis aligned)
    assert torch.allclose(metrics.g_mpjpe(p, p), torch.zeros(4, 3, device=dev))
    assert metrics.std_diversity(p[..., :24, :]).shape == (4,)


@pytest.mark.gpu
def test_diversity_metrics_match_reference_loops():
    """std / apd diversity (ehm_eval_diversity) incl. the visible / invisible joint splits against a literal restatement of the per-item numpy
    loops of test_egohmr.py:453-494."""
    import numpy as np
    import torch
    from egohmr_amd import metrics
    dev = torch.device("cuda:0")
    g = np.random.Generator(np.random.PCG64(7))
    B, S = 5, 4
    a = g.normal(size=(B, S, 24, 3)).astype(np.float32)
    mask = g.random((B, 24)) < 0.6
    mask[0] = True            # nothing invisible for item 0 -> NaN in the invisible split, like the reference
    t, m = torch.from_numpy(a).to(dev), torch.from_numpy(mask).to(dev)
    # reference-style loops
    std_all = a.std(axis=1, ddof=1).mean(-1).mean(-1)
    pd = np.linalg.norm(a[:, None] - a[:, :, None], axis=-1)
    apd_all = pd.sum(axis=(-1, -2, -3)) / 24 / S / (S - 1) / 2
    np.testing.assert_allclose(metrics.std_diversity(t).cpu().numpy(), std_all, rtol=1e-5)
    np.testing.assert_allclose(metrics.apd_diversity(t).cpu().numpy(), apd_all, rtol=1e-5)
    for sel in (mask, ~mask):
        std_ref, apd_ref = [], []
        for k in range(B):
            tmp = a[k][:, sel[k]]
            with np.errstate(invalid="ignore", divide="ignore"):
                std_ref.append(tmp.std(axis=0, ddof=1).mean(-1).mean(-1) if tmp.shape[1] else np.nan)
                d = np.linalg.norm(tmp[None] - tmp[:, None], axis=-1)
                apd_ref.append(d.sum() / tmp.shape[-2] / S / (S - 1) / 2 if tmp.shape[1] else np.nan)
        np.testing.assert_allclose(metrics.std_diversity_masked(t, torch.from_numpy(sel).to(dev)).cpu().numpy(), np.array(std_ref, np.float32), rtol=1e-5)
        np.testing.assert_allclose(metrics.apd_diversity(t, torch.from_numpy(sel).to(dev)).cpu().numpy(), np.array(apd_ref, np.float32), rtol=1e-5)
    sd_inv, apd_inv = metrics.diversity(t, m, invert=True)          # the inverted selection inside the kernel = the complement mask
    sd_c, apd_c = metrics.diversity(t, ~m)
    assert torch.equal(torch.nan_to_num(sd_inv, nan=-1.0), torch.nan_to_num(sd_c, nan=-1.0)) and torch.equal(torch.nan_to_num(apd_inv, nan=-1.0), torch.nan_to_num(apd_c, nan=-1.0))


@pytest.mark.gpu
def test_point_errors_kernel_vs_numpy():
    """ehm_eval_point_errors: G-MPJPE / MPJPE / V2V with their visible / invisible sums (test_egohmr.py:399-449) against float64 numpy: S samples against one
    ground truth per item, 45 model joints of which 24 count, 6890 vertices (27 rounds of the block), origins by pointer and by index."""
    from egohmr_amd import metrics
    dev = torch.device("cuda:0")
    g = np.random.Generator(np.random.PCG64(5))
    B, S = 3, 4
    pj, gj = g.normal(size=(B, S, 45, 3)).astype(np.float32), g.normal(size=(B, 24, 3)).astype(np.float32)
    jm = g.random((B, 24)) < 0.5
    r = metrics.point_errors(torch.from_numpy(pj).to(dev), torch.from_numpy(gj).to(dev), points=24, mask=torch.from_numpy(jm).to(dev), per_point=True)
    e = np.sqrt(((pj[:, :, :24].astype(np.float64) - gj[:, None]) ** 2).sum(-1))
    np.testing.assert_allclose(r["per_point"].cpu().numpy(), e, rtol=1e-6)
    np.testing.assert_allclose(r["mean"].cpu().numpy(), e.mean(-1), rtol=1e-5)
    np.testing.assert_allclose(r["vis_sum"].cpu().numpy(), (e * jm[:, None]).sum(-1), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(r["invis_sum"].cpu().numpy(), (e * ~jm[:, None]).sum(-1), rtol=1e-5, atol=1e-6)
    r0 = metrics.point_errors(torch.from_numpy(pj).to(dev), torch.from_numpy(gj).to(dev), points=24, origin_point=0)
    e0 = np.sqrt((((pj[:, :, :24] - pj[:, :, :1]).astype(np.float64) - (gj - gj[:, :1])[:, None]) ** 2).sum(-1))
    np.testing.assert_allclose(r0["mean"].cpu().numpy(), e0.mean(-1), rtol=1e-5)
    np.testing.assert_allclose(r0["vis_sum"].cpu().numpy(), e0.sum(-1), rtol=1e-5)            # no mask: everything is visible
    assert float(r0["invis_sum"].abs().max()) == 0.0
    pv, gv = g.normal(size=(B, S, 6890, 3)).astype(np.float32), g.normal(size=(B, 6890, 3)).astype(np.float32)
    po, go = g.normal(size=(B, S, 1, 3)).astype(np.float32), g.normal(size=(B, 1, 3)).astype(np.float32)
    vm = g.random((B, 6890)) < 0.7
    ev = np.sqrt((((pv - po).astype(np.float64) - (gv - go)[:, None]) ** 2).sum(-1))
    rv = metrics.point_errors(torch.from_numpy(pv).to(dev), torch.from_numpy(gv).to(dev), pred_origin=torch.from_numpy(po).to(dev), gt_origin=torch.from_numpy(go).to(dev),
                              mask=torch.from_numpy(vm).to(dev))
    np.testing.assert_allclose(rv["mean"].cpu().numpy(), ev.mean(-1), rtol=2e-5)
    np.testing.assert_allclose(rv["vis_sum"].cpu().numpy(), (ev * vm[:, None]).sum(-1), rtol=2e-5)
    np.testing.assert_allclose(metrics.v2v(torch.from_numpy(pv).to(dev), torch.from_numpy(po).to(dev), torch.from_numpy(gv).unsqueeze(1).to(dev),
                                           torch.from_numpy(go).unsqueeze(1).to(dev)).cpu().numpy(), ev.mean(-1), rtol=2e-5)
    with pytest.raises(Exception):
        metrics.mpjpe(torch.from_numpy(pj), torch.from_numpy(gj))                             # CPU tensors: no CPU path


@pytest.mark.gpu
def test_procrustes_kernel_vs_oracle_edge_cases(golden_dir):
    """ehm_eval_procrustes (one-sided Jacobi SVD in float64 registers) against the numpy restatement of utils/pose_utils.py:10-66 (itself pinned by the
    reference golden G11): the golden's pairs with S samples per ground truth, reflected clouds (det(U V^T) < 0: Z flips the smallest singular direction),
    planar clouds (rank-2 K), scaled / rotated copies (error 0), per-joint errors and masked sums."""
    from egohmr_amd import metrics
    dev = torch.device("cuda:0")
    g11 = np.load(os.path.join(golden_dir, "g11_procrustes.npz"))
    gt = g11["gt"].astype(np.float32)                                    # [n,J,3]
    n, J = gt.shape[:2]
    rng = np.random.Generator(np.random.PCG64(2))
    S = 5
    pred = np.stack([g11["pred"].astype(np.float32)] + [gt + 0.1 * rng.normal(size=gt.shape).astype(np.float32) for _ in range(S - 1)], 1)   # [n,S,J,3]
    pred[:, 2] = pred[:, 2] * np.array([1, 1, -1], np.float32)           # a mirrored body: the best ROTATION needs the sign fix
    pred[:, 3, :, 2] = 0.0                                                # a planar prediction: rank-2 correlation matrix
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    q *= np.sign(np.linalg.det(q))
    pred[:, 4] = (2.5 * gt @ q.T + np.array([0.3, -1.0, 2.0])).astype(np.float32)   # a similarity transform of the truth: error 0
    mask = rng.random((n, J)) < 0.5
    r = metrics.procrustes(torch.from_numpy(pred).to(dev), torch.from_numpy(gt).to(dev), mask=torch.from_numpy(mask).to(dev), aligned=True, per_joint=True)
    want = np.stack([np.stack([om.procrustes(pred[i, s].astype(np.float64), gt[i].astype(np.float64)) for s in range(S)]) for i in range(n)])
    e = np.sqrt(((want - gt[:, None]) ** 2).sum(-1))
    np.testing.assert_allclose(r["aligned"].cpu().numpy(), want, atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(r["per_joint"].cpu().numpy(), e, atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(r["mean"].cpu().numpy(), e.mean(-1), atol=1e-6, rtol=1e-5)
    np.testing.assert_allclose(r["vis_sum"].cpu().numpy(), (e * mask[:, None]).sum(-1), atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(r["invis_sum"].cpu().numpy(), (e * ~mask[:, None]).sum(-1), atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(r["mean"][:, 0].cpu().numpy(), g11["pa_mpjpe"], rtol=2e-5, atol=1e-6)       # the reference's own numbers
    assert float(r["mean"][:, 4].max()) < 1e-5
    al = metrics.similarity_align(torch.from_numpy(pred[:, 1]).to(dev), torch.from_numpy(gt).to(dev))
    np.testing.assert_allclose(al.cpu().numpy(), want[:, 1], atol=2e-6, rtol=1e-5)


def test_results_wire_format_roundtrip(tmp_path):
    """results_seed_*.pkl: the reference's keys, numpy payloads, pickle protocol 2 (test_egohmr.py:672-695); stage-1 cam file;
    preprocess stats / mean-params loaders and their shape checks."""
    import pickle
    import numpy as np
    import pytest
    import torch
    from egohmr_amd import io as eio
    n, S = 6, 3
    g = np.random.Generator(np.random.PCG64(3))
    res = eio.results_dict(torch.from_numpy(g.normal(size=(n, S, 10)).astype(np.float32)), g.normal(size=(n, S, 1, 3, 3)).astype(np.float32),
                           g.normal(size=(n, S, 23, 3, 3)).astype(np.float32), g.random((n, S)), g.random((n, S)),
                           g.normal(size=(n, 3)).astype(np.float32))
    assert list(res.keys()) == ["pred_betas_list", "pred_global_orient_list", "pred_body_pose_list", "collision_ratio_list",
                                "contact_ratio_list", "gt_cam_full_list"]
    path = eio.save_results(str(tmp_path), "53618", 0, res)
    assert path.endswith("output_egohmr_53618/results_seed_0.pkl")
    raw = open(path, "rb").read()
    assert raw[:2] == b"\x80\x02"                               # protocol 2 header
    back = eio.load_results(path)
    for k, v in res.items():
        assert isinstance(back[k], np.ndarray)
        np.testing.assert_array_equal(back[k], v)
    with pytest.raises(KeyError):
        eio.save_results(str(tmp_path), "x", 1, {"pred_betas_list": res["pred_betas_list"]})
    # two-stage: the stage-1 file and the extra key
    s1 = tmp_path / "results.pkl"
    with open(s1, "wb") as f:
        pickle.dump({"pred_cam_full_list": g.normal(size=(n, 3))}, f, protocol=2)
    cam = eio.load_stage1_cam(str(s1))
    assert cam.shape == (n, 3) and cam.dtype == np.float32
    res2 = eio.results_dict(res["pred_betas_list"], res["pred_global_orient_list"], res["pred_body_pose_list"], res["collision_ratio_list"],
                            res["contact_ratio_list"], res["gt_cam_full_list"], pred_cam_full=cam)
    assert list(res2.keys())[-2:] == ["pred_cam_full_list", "gt_cam_full_list"]   # the reference's insertion order
    with open(tmp_path / "bad.pkl", "wb") as f:
        pickle.dump({"something": 1}, f, protocol=2)
    with pytest.raises(KeyError):
        eio.load_stage1_cam(str(tmp_path / "bad.pkl"))
    # statistics files
    np.savez(tmp_path / "preprocess_stats.npz", Xmean=np.arange(144, dtype=np.float64), Xstd=np.ones(144))
    mean, std = eio.load_preprocess_stats(str(tmp_path / "preprocess_stats.npz"))
    assert mean.dtype == np.float32 and mean.shape == (144,) and std.shape == (144,)
    np.savez(tmp_path / "smpl_mean_params.npz", shape=np.arange(10, dtype=np.float64), pose=np.zeros(144), cam=np.zeros(3))
    assert eio.load_smpl_mean_params(str(tmp_path / "smpl_mean_params.npz")).shape == (1, 10)
    np.savez(tmp_path / "short.npz", Xmean=np.zeros(10), Xstd=np.zeros(10))
    with pytest.raises(ValueError):
        eio.load_preprocess_stats(str(tmp_path / "short.npz"))
