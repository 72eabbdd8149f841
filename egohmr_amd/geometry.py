"""Rotation-representation helpers with the reference's names (utils/geometry.py).

``rot6d_to_rotmat`` runs the gfx950 kernel (csrc/smpl.hip: rot6d_kernel) - it is on the hot path
(utils/geometry.py:47-66, called at models/egohmr/egohmr.py:260 and :529).  The remaining helpers
are off the per-step path and are thin torch expressions kept for API completeness.
"""
from __future__ import annotations

import torch

from . import _lib

_MODES = {"prohmr": 0, "diffusion": 1}


class _Rot6dToRotmat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mode):
        x = _lib.f32(x).reshape(-1, 6)
        n = x.shape[0]
        R = torch.empty(n, 3, 3, device=x.device, dtype=torch.float32)
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().ehm_rot6d_to_rotmat(_lib.ptr(x), _lib.ptr(R), n, mode, _lib.stream_ptr()), "ehm_rot6d_to_rotmat")
        ctx.save_for_backward(x)
        ctx.mode = mode
        return R

    @staticmethod
    def backward(ctx, gR):
        (x,) = ctx.saved_tensors
        gR = _lib.f32(gR)
        gx = torch.empty_like(x)
        with _lib.on_device(x.device):
            _lib.check(_lib.lib().ehm_rot6d_to_rotmat_bwd(_lib.ptr(x), _lib.ptr(gR), _lib.ptr(gx), x.shape[0], ctx.mode,
                                                          _lib.stream_ptr()), "ehm_rot6d_to_rotmat_bwd")
        return gx, None


def rot6d_to_rotmat(x: torch.Tensor, rot6d_mode: str = "prohmr") -> torch.Tensor:
    """utils/geometry.py:47-66.  x: any shape with 6*n elements -> [n,3,3]; autograd-capable."""
    if rot6d_mode not in _MODES:
        raise ValueError(f"unknown rot6d_mode {rot6d_mode!r}")
    return _Rot6dToRotmat.apply(x, _MODES[rot6d_mode])


def rotmat_to_rot6d(x_batch: torch.Tensor, rot6d_mode: str = "prohmr") -> torch.Tensor:
    """utils/geometry.py:69-75 (only the 'diffusion' layout is defined by the reference)."""
    if rot6d_mode != "diffusion":
        raise NotImplementedError("the reference leaves rot6d_mode != 'diffusion' undefined (utils/geometry.py:73-74)")
    return x_batch[:, :, :-1].reshape(-1, 6)


def aa_to_rotmat(theta: torch.Tensor) -> torch.Tensor:
    """utils/geometry.py:5-45 (training / data preparation only)."""
    angle = torch.norm(theta + 1e-8, p=2, dim=1, keepdim=True)
    axis = theta / angle
    q = torch.cat([torch.cos(angle * 0.5), torch.sin(angle * 0.5) * axis], dim=1)
    q = q / q.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = q.unbind(dim=1)
    return torch.stack([w * w + x * x - y * y - z * z, 2 * x * y - 2 * w * z, 2 * w * y + 2 * x * z,
                        2 * w * z + 2 * x * y, w * w - x * x + y * y - z * z, 2 * y * z - 2 * w * x,
                        2 * x * z - 2 * w * y, 2 * w * x + 2 * y * z, w * w - x * x - y * y + z * z], dim=1).view(-1, 3, 3)


def perspective_projection(points, translation, focal_length, camera_center=None, rotation=None):
    """utils/geometry.py:78-116 (output garnish of EgoHMR.forward, egohmr.py:294-301)."""
    if rotation is not None:
        points = torch.einsum("bij,bkj->bki", rotation, points)
    p = points + translation.unsqueeze(1)
    p = p / p[:, :, -1:]
    if camera_center is None:
        camera_center = torch.zeros_like(focal_length)
    u = focal_length[:, None, 0] * p[:, :, 0] + camera_center[:, None, 0] * p[:, :, 2]
    v = focal_length[:, None, 1] * p[:, :, 1] + camera_center[:, None, 1] * p[:, :, 2]
    return torch.stack((u, v), dim=-1)


def _safe_div(num, den, eps=1.0e-6):
    den = den.clone()
    den[den.abs() < eps] += eps
    den[den.abs() < eps] += eps       # applied twice, as utils/konia_transform.py:343-347 does
    return num / den


def rotation_matrix_to_angle_axis(rotation_matrix: torch.Tensor) -> torch.Tensor:
    """utils/konia_transform.py:316-340 (-> rotation_matrix_to_quaternion :349-443, quaternion_to_angle_axis :560-630).
    Only feeds ``full_pose`` of COAP-style collision models (egohmr.py:495,540); the build's proxy does not consume it,
    so this stays a torch expression off the per-step path."""
    if rotation_matrix.shape[-2:] != (3, 3):
        raise ValueError(f"Input size must be a (*, 3, 3) tensor. Got {rotation_matrix.shape}")
    m = rotation_matrix.reshape(*rotation_matrix.shape[:-2], 9)
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.chunk(m, 9, dim=-1)
    trace = m00 + m11 + m22
    eps = 1.0e-6

    def branch(lead, w_num, x_num, y_num, z_num, order):
        sq = torch.sqrt(lead.clamp_min(eps)) * 2.0
        parts = {"lead": 0.25 * sq, "w": _safe_div(w_num, sq) if w_num is not None else None,
                 "x": _safe_div(x_num, sq) if x_num is not None else None, "y": _safe_div(y_num, sq) if y_num is not None else None,
                 "z": _safe_div(z_num, sq) if z_num is not None else None}
        return torch.cat([parts["lead"] if o == order else parts[o] for o in "wxyz"], dim=-1)

    q_pos = branch(trace + 1.0, None, m21 - m12, m02 - m20, m10 - m01, "w")
    q_1 = branch(1.0 + m00 - m11 - m22, m21 - m12, None, m01 + m10, m02 + m20, "x")
    q_2 = branch(1.0 + m11 - m00 - m22, m02 - m20, m01 + m10, None, m12 + m21, "y")
    q_3 = branch(1.0 + m22 - m00 - m11, m10 - m01, m02 + m20, m12 + m21, None, "z")
    q = torch.where(trace > 0.0, q_pos, torch.where((m00 > m11) & (m00 > m22), q_1, torch.where(m11 > m22, q_2, q_3)))
    cos_t, q1, q2, q3 = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    s2 = q1 * q1 + q2 * q2 + q3 * q3
    sn = torch.sqrt(s2.clamp_min(eps))

    def safe_atan2(y, x):
        y = y.clone()
        y[(y.abs() < eps) & (x.abs() < eps)] += eps
        return torch.atan2(y, x)

    two_theta = 2.0 * torch.where(cos_t < 0.0, safe_atan2(-sn, -cos_t), safe_atan2(sn, cos_t))
    k = torch.where(s2 > 0.0, _safe_div(two_theta, sn, eps), 2.0 * torch.ones_like(sn))
    return torch.stack((q1 * k, q2 * k, q3 * k), dim=-1)
