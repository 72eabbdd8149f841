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
