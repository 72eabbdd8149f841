"""Oracle: the build-defined collision proxy (CPU, autograd).  Test infrastructure only.

NOT a restatement of reference code.  The reference calls ``smpl.coap.collision_loss`` (COAP,
models/egohmr/egohmr.py:555) or ``smpl.volume.collision_loss`` (VolumetricSMPL,
models/egohmr/egohmr_volsmpl.py:612): learned occupancy / SDF networks whose code and weights are
not in /root/reference and cannot be fetched offline - PARITY UNPINNED at that boundary.  The
north-star names a "scene-point Chamfer/SDF guidance reduction"; this proxy is its definition here:

    d_p   = min_v || p - v ||           (scene point p already bbox-selected; v body vertices)
    loss  = sum_p  relu(tau - d_p)^2    (tau = 5 cm: scene points nearer than tau count as contact)

The gradient flows to the arg-min vertex only, like ``torch.min``.
"""
from __future__ import annotations

import torch

TAU = 0.05


def proxy_collision_loss(points, verts, joints=None, full_pose_aa=None, tau: float = TAU):
    """points [1,n,3], verts [1,V,3] -> scalar."""
    p = points[0]
    v = verts[0]
    total = torch.zeros((), dtype=v.dtype)
    for s in range(0, p.shape[0], 1024):
        diff = p[s:s + 1024, None, :] - v[None, :, :]
        d2 = (diff * diff).sum(-1)
        d2min, _ = d2.min(dim=1)
        d = torch.sqrt(d2min + 1e-12)
        total = total + (torch.relu(tau - d) ** 2).sum()
    return total


def proxy_min_dist(points, verts):
    """points [n,3], verts [V,3] -> distance of every point to its nearest vertex (same epsilon as the loss)."""
    out = []
    for s in range(0, points.shape[0], 1024):
        diff = points[s:s + 1024, None, :] - verts[None, :, :]
        out.append(torch.sqrt((diff * diff).sum(-1).min(dim=1).values + 1e-12))
    return torch.cat(out) if out else points.new_zeros(0)


def proxy_collision_loss_batched(points, verts, tau: float = TAU):
    """The VolSMPL-shaped call `volume.collision_loss(points [B,N,3], smpl_output) -> [B]` (models/egohmr/egohmr_volsmpl.py:609-612):
    every scene point of every item, no bounding-box selection."""
    return torch.stack([proxy_collision_loss(points[[b]], verts[[b]], tau=tau) for b in range(points.shape[0])])


def proxy_occupancy(points, verts, tau: float = TAU):
    """Stand-in for `coap.query(points [1,n,3], smpl_output) -> occupancy [1,n]` (egohmr.py:509, `> 0.5` = inside):
    1 where the point is closer than tau to the surface, else 0."""
    return (proxy_min_dist(points[0], verts[0]) < tau).to(verts.dtype).unsqueeze(0)


def proxy_sdf(points, verts, tau: float = TAU):
    """Stand-in for `volume.query_fast(points [1,n,3], smpl_output) -> sdf [1,n]` (egohmr_volsmpl.py:574, `< 0` = inside)."""
    return (proxy_min_dist(points[0], verts[0]) - tau).unsqueeze(0)
