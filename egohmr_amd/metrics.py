"""Evaluation metrics of the reference's driver, on the device (SURVEY.md section 8f "next" rows 1 and 3).

* chamfer_distance / contact_score   utils/pytorch3d_chamfer_distance.py:70-218, test_egohmr.py:496-505
  (nearest-neighbour search = HIP kernel ehm_nn_dist2; the reference uses pytorch3d's CUDA knn_points)
* mpjpe / g_mpjpe / v2v                test_egohmr.py:399-443           -> ehm_eval_point_errors (csrc/eval.hip)
* pa_mpjpe (batched Procrustes)        utils/pose_utils.py:10-66, :109-126 (the reference loops numpy SVDs per sample on the CPU)
                                       -> ehm_eval_procrustes: float64 in registers, 3 x 3 SVD by one-sided Jacobi, one thread per sample
* std_diversity / apd                  test_egohmr.py:453-494           -> ehm_eval_diversity (one wave per item)
Every function here is a launch of a hand-written kernel; CPU tensors raise (the CPU restatement is oracle/metrics.py, test infrastructure).
"""
from __future__ import annotations

import torch

from . import _lib


def nn_dist2(x: torch.Tensor, y: torch.Tensor, return_idx: bool = False):
    """Squared distance from every x[b,i] to its nearest y[b,:] (pytorch3d knn_points K=1 'dists')."""
    x, y = _lib.f32(x), _lib.f32(y)
    assert x.dim() == 3 and y.dim() == 3 and x.shape[0] == y.shape[0] and x.shape[2] == y.shape[2] == 3
    if not x.is_cuda:
        raise _lib.EgoHMRHipError("egohmr_amd.metrics runs on the HIP kernels of csrc/metrics.hip: tensors must live on a HIP device (there is no CPU path)")
    B, P1, P2 = x.shape[0], x.shape[1], y.shape[1]
    d = torch.empty(B, P1, device=x.device, dtype=torch.float32)
    idx = torch.empty(B, P1, device=x.device, dtype=torch.int32) if return_idx else None
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().ehm_nn_dist2(_lib.ptr(x), _lib.ptr(y), _lib.ptr(d), _lib.ptr(idx), B, P1, P2, _lib.stream_ptr()), "ehm_nn_dist2")
    return (d, idx) if return_idx else d


def chamfer_distance(x, y, x_lengths=None, y_lengths=None, x_normals=None, y_normals=None, weights=None,
                     batch_reduction=None, point_reduction="mean"):
    """The reference's (modified) chamfer_distance: returns per-point squared NN distances
    (cham_x [N,P1], cham_y [N,P2], None) - its point reduction is commented out (pytorch3d_chamfer_distance.py:186-195)."""
    if x_lengths is not None or y_lengths is not None or x_normals is not None or y_normals is not None:
        raise NotImplementedError("ragged clouds / normals are not used by the EgoHMR driver")
    cham_x, cham_y = nn_dist2(x, y), nn_dist2(y, x)
    if weights is not None:
        cham_x, cham_y = cham_x * weights.view(-1, 1), cham_y * weights.view(-1, 1)
    if batch_reduction is not None:
        cham_x, cham_y = cham_x.sum(), cham_y.sum()
        if batch_reduction == "mean":
            div = weights.sum() if weights is not None else x.shape[0]
            cham_x, cham_y = cham_x / div, cham_y / div
    return cham_x, cham_y, None


def contact_score(pred_vertices_full: torch.Tensor, scene_pcd: torch.Tensor, thres: float = 0.02) -> torch.Tensor:
    """test_egohmr.py:496-505: a body is 'in contact' when its minimum SQUARED vertex-scene distance is < thres."""
    return nn_dist2(pred_vertices_full, scene_pcd).min(dim=-1)[0] < thres


def _f32c(t):
    if not t.is_cuda:
        raise _lib.EgoHMRHipError("egohmr_amd.metrics runs on the HIP kernels of csrc/eval.hip: tensors must live on a HIP device (there is no CPU path)")
    return _lib.f32(t)


def _mask_u8(mask, B, P, device):
    if mask is None:
        return None
    m = mask.to(device=device, dtype=torch.uint8).contiguous()
    assert m.shape == (B, P), (tuple(m.shape), (B, P))
    return m


def _bs_form(pred, gt):
    """(pred [B,S,P',3], gt [B,P'',3], leading shape of the result) for pred [B,S,P',3] against gt [B,1,P'',3] / [B,P'',3] (S samples per item, one ground
    truth per item: test_egohmr.py's gt.unsqueeze(1)), or for pred and gt with the SAME leading shape (every row its own ground truth)."""
    import math
    lead = pred.shape[:-2]
    if pred.dim() == 4 and gt.dim() == 4 and gt.shape[0] == pred.shape[0] and gt.shape[1] == 1:
        gt = gt[:, 0]
    if pred.dim() == 4 and gt.dim() == 3 and gt.shape[0] == pred.shape[0]:
        return pred, gt, lead
    assert gt.shape[:-2] == lead, (tuple(pred.shape), tuple(gt.shape))
    n = math.prod(lead)
    return pred.reshape(n, 1, *pred.shape[-2:]), gt.reshape(n, *gt.shape[-2:]), lead


def point_errors(pred: torch.Tensor, gt: torch.Tensor, points: int | None = None, origin_point: int = -1, pred_origin=None, gt_origin=None, mask=None,
                 per_point: bool = False) -> dict:
    """ehm_eval_point_errors (csrc/eval.hip): per-point Euclidean error of pred [B,S,P',3] against gt [B,P'',3] (first `points` of each), mean over the
    points and sums over the visible / invisible ones (mask [B,P]) - the reductions of test_egohmr.py:399-449 in one launch."""
    pred, gt = _f32c(pred), _f32c(gt)
    B, S = pred.shape[0], pred.shape[1]
    P = int(points) if points is not None else min(pred.shape[2], gt.shape[1])
    dev = pred.device
    mean = torch.empty(B, S, device=dev)
    vis, invis = torch.empty(B, S, device=dev), torch.empty(B, S, device=dev)
    pp = torch.empty(B, S, P, device=dev) if per_point else None
    m = _mask_u8(mask, B, P, dev)
    po = _f32c(pred_origin).reshape(B, S, 3) if pred_origin is not None else None
    go = _f32c(gt_origin).reshape(B, 3) if gt_origin is not None else None
    d = _lib.EvalPointsDesc(_lib.ptr(pred), _lib.ptr(gt), _lib.ptr(po), _lib.ptr(go), _lib.ptr(m), _lib.ptr(pp), _lib.ptr(mean), _lib.ptr(vis), _lib.ptr(invis),
                            B, S, P, pred.shape[2], gt.shape[1], int(origin_point))
    import ctypes as C
    with _lib.on_device(dev):
        _lib.check(_lib.lib().ehm_eval_point_errors(C.byref(d), _lib.stream_ptr()), "ehm_eval_point_errors")
    return {"mean": mean, "vis_sum": vis, "invis_sum": invis, "per_point": pp}


def mpjpe(pred_joints: torch.Tensor, gt_joints: torch.Tensor) -> torch.Tensor:
    """test_egohmr.py:409-411: pelvis-aligned mean per-joint error over the first 24 joints -> [...]."""
    p, g, lead = _bs_form(pred_joints, gt_joints)
    return point_errors(p, g, points=24, origin_point=0)["mean"].reshape(lead)


def g_mpjpe(pred_joints_full: torch.Tensor, gt_joints_full: torch.Tensor) -> torch.Tensor:
    """test_egohmr.py:399-401: error in the camera frame, no alignment."""
    p, g, lead = _bs_form(pred_joints_full, gt_joints_full)
    return point_errors(p, g, points=24)["mean"].reshape(lead)


def v2v(pred_vertices: torch.Tensor, pred_pelvis, gt_vertices: torch.Tensor, gt_pelvis) -> torch.Tensor:
    """test_egohmr.py:441-443: pelvis-aligned mean vertex-to-vertex error."""
    p, g, lead = _bs_form(pred_vertices, gt_vertices)
    B, S = p.shape[0], p.shape[1]
    po = pred_pelvis.reshape(-1, 3)
    po = (po if po.shape[0] == B * S else po.reshape(B, -1, 3).expand(B, S, 3)).reshape(B, S, 3)      # [..., 1, 3] per sample, or one pelvis per item
    go = gt_pelvis.reshape(-1, 3)
    go = go if go.shape[0] == B else go.expand(B, 3)
    return point_errors(p, g, pred_origin=po, gt_origin=go)["mean"].reshape(lead)


def procrustes(pred: torch.Tensor, gt: torch.Tensor, mask=None, aligned: bool = False, per_joint: bool = False) -> dict:
    """ehm_eval_procrustes (csrc/eval.hip): utils/pose_utils.py:10-66 batched - pred [B,S,J,3] against gt [B,J,3], float64 inside the kernel."""
    pred, gt = _f32c(pred), _f32c(gt)
    B, S, J = pred.shape[0], pred.shape[1], pred.shape[2]
    assert gt.shape == (B, J, 3), (tuple(pred.shape), tuple(gt.shape))
    dev = pred.device
    al = torch.empty(B, S, J, 3, device=dev) if aligned else None
    pj = torch.empty(B, S, J, device=dev) if per_joint else None
    mean, vis, invis = torch.empty(B, S, device=dev), torch.empty(B, S, device=dev), torch.empty(B, S, device=dev)
    m = _mask_u8(mask, B, J, dev)
    with _lib.on_device(dev):
        _lib.check(_lib.lib().ehm_eval_procrustes(_lib.ptr(pred), _lib.ptr(gt), _lib.ptr(m), _lib.ptr(al), _lib.ptr(pj), _lib.ptr(mean), _lib.ptr(vis), _lib.ptr(invis),
                                                  B, S, J, _lib.stream_ptr()), "ehm_eval_procrustes")
    return {"aligned": al, "per_joint": pj, "mean": mean, "vis_sum": vis, "invis_sum": invis}


def similarity_align(S1: torch.Tensor, S2: torch.Tensor) -> torch.Tensor:
    """utils/pose_utils.py:10-66 batched on the device: the similarity transform of S1 [n,J,3] closest to S2 [n,J,3] (float32 out, float64 inside)."""
    return procrustes(S1.reshape(-1, 1, *S1.shape[-2:]), S2.reshape(-1, *S2.shape[-2:]), aligned=True)["aligned"].reshape(S1.shape)


def pa_mpjpe(pred_joints: torch.Tensor, gt_joints: torch.Tensor) -> torch.Tensor:
    """utils/pose_utils.py:109-126 (reconstruction_error): Procrustes-aligned mean joint error -> [n]."""
    p = pred_joints.reshape(-1, 1, pred_joints.shape[-2], 3)
    g = gt_joints.reshape(-1, gt_joints.shape[-2], 3)
    return procrustes(p, g)["mean"].reshape(-1)


def diversity(pred_joints_aligned: torch.Tensor, joint_mask=None, invert: bool = False):
    """ehm_eval_diversity (csrc/eval.hip): (std, apd) [B] of joints [B,S,J,3] over the joints selected by joint_mask [B,J] (None: all; invert: the others)."""
    a = _f32c(pred_joints_aligned)
    B, S, J = a.shape[0], a.shape[1], a.shape[2]
    sd, apd = torch.empty(B, device=a.device), torch.empty(B, device=a.device)
    m = _mask_u8(joint_mask, B, J, a.device)
    with _lib.on_device(a.device):
        _lib.check(_lib.lib().ehm_eval_diversity(_lib.ptr(a), _lib.ptr(m), int(bool(invert)), _lib.ptr(sd), _lib.ptr(apd), B, S, J, _lib.stream_ptr()), "ehm_eval_diversity")
    return sd, apd


def std_diversity(pred_joints_aligned: torch.Tensor) -> torch.Tensor:
    """test_egohmr.py:453-455: std over the sample axis of [B,S,24,3], averaged over joints and coordinates."""
    return diversity(pred_joints_aligned)[0]


def std_diversity_masked(pred_joints_aligned: torch.Tensor, joint_mask: torch.Tensor) -> torch.Tensor:
    """test_egohmr.py:457-470 (std-joints-vis / -invis): the per-item loop `pred[k, :, mask[k]]` batched - std over samples,
    mean over the selected joints and the 3 coordinates.  Items without a selected joint give NaN like the reference."""
    return diversity(pred_joints_aligned, joint_mask)[0]


def apd_diversity(pred_joints_aligned: torch.Tensor, joint_mask: torch.Tensor | None = None) -> torch.Tensor:
    """test_egohmr.py:471-494 (apd-joints, -vis, -invis): sum over ordered sample pairs and (selected) joints of the joint
    distance, divided by n_joints * S * (S-1) * 2 - the reference's normalisation, kept as is."""
    return diversity(pred_joints_aligned, joint_mask)[1]
