"""Oracle: evaluation metrics on the CPU (numpy).  Test infrastructure only.

nn_dist2 restates what pytorch3d.ops.knn_points(K=1) returns (squared distance to the nearest neighbour;
utils/pytorch3d_chamfer_distance.py:152-156 - pytorch3d 0.6.1 is a CUDA extension absent from the reference tree, so this
row is pinned by definition only); procrustes / pa_mpjpe follow utils/pose_utils.py:10-66,109-126 and are pinned by
tests/golden/g11_procrustes.npz generated from the reference itself.
"""
import numpy as np


def nn_dist2(x, y):
    """x [B,P1,3], y [B,P2,3] -> (dist2 [B,P1], idx [B,P1])."""
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    d = np.empty(x.shape[:2])
    idx = np.empty(x.shape[:2], np.int64)
    for b in range(x.shape[0]):
        for s in range(0, x.shape[1], 512):
            diff = x[b, s:s + 512, None, :] - y[b, None, :, :]
            d2 = (diff * diff).sum(-1)
            idx[b, s:s + 512] = d2.argmin(1)
            d[b, s:s + 512] = d2.min(1)
    return d, idx


def procrustes(S1, S2):
    """utils/pose_utils.py:10-66 for one [J,3] pair."""
    S1, S2 = S1.T, S2.T
    mu1, mu2 = S1.mean(axis=1, keepdims=True), S2.mean(axis=1, keepdims=True)
    X1, X2 = S1 - mu1, S2 - mu2
    var1 = np.sum(X1 ** 2)
    K = X1.dot(X2.T)
    U, s, Vh = np.linalg.svd(K)
    V = Vh.T
    Z = np.eye(3)
    Z[-1, -1] *= np.sign(np.linalg.det(U.dot(V.T)))
    R = V.dot(Z.dot(U.T))
    scale = np.trace(R.dot(K)) / var1
    t = mu2 - scale * (R.dot(mu1))
    return (scale * R.dot(S1) + t).T


def pa_mpjpe(S1, S2):
    """utils/pose_utils.py:109-116 with avg_joint=True."""
    hat = np.stack([procrustes(a, b) for a, b in zip(S1, S2)])
    return np.sqrt(((hat - S2) ** 2).sum(-1)).mean(-1)
