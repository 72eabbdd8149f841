"""Oracle: rotation-representation maths on the CPU (torch, any float dtype).  Test infrastructure only.

rot6d_to_rotmat / rotmat_to_rot6d / aa_to_rotmat / perspective_projection follow
utils/geometry.py:5-116; rotation_matrix_to_angle_axis follows utils/konia_transform.py:316-340
(-> rotation_matrix_to_quaternion :349-443, quaternion_to_angle_axis :560-630,
safe_zero_division :343-347, torch_safe_atan2 :44-47).
"""
from __future__ import annotations

import torch


def rot6d_to_rotmat(x: torch.Tensor, rot6d_mode: str = "prohmr") -> torch.Tensor:
    """geometry.py:47-66.  'diffusion' layout: a1 = x[0::2], a2 = x[1::2] of each 6-vector."""
    if rot6d_mode == "prohmr":
        x = x.reshape(-1, 2, 3).permute(0, 2, 1)
    elif rot6d_mode == "diffusion":
        x = x.reshape(-1, 3, 2)
    else:
        raise ValueError(rot6d_mode)
    a1, a2 = x[:, :, 0], x[:, :, 1]
    b1 = a1 / a1.norm(dim=1, keepdim=True).clamp_min(1e-12)          # F.normalize, eps=1e-12
    u = a2 - (b1 * a2).sum(dim=1, keepdim=True) * b1
    b2 = u / u.norm(dim=1, keepdim=True).clamp_min(1e-12)
    b3 = torch.linalg.cross(b1, b2, dim=1)
    return torch.stack((b1, b2, b3), dim=-1)


def rotmat_to_rot6d(R: torch.Tensor, rot6d_mode: str = "diffusion") -> torch.Tensor:
    """geometry.py:69-75: first two columns, row-major -> [r00,r01,r10,r11,r20,r21]."""
    if rot6d_mode != "diffusion":
        raise NotImplementedError("reference leaves the 'prohmr' branch unimplemented (geometry.py:73-74)")
    return R[:, :, :-1].reshape(-1, 6)


def aa_to_rotmat(theta: torch.Tensor) -> torch.Tensor:
    """geometry.py:5-45 (axis-angle -> quaternion -> matrix)."""
    angle = torch.norm(theta + 1e-8, p=2, dim=1).unsqueeze(-1)
    axis = theta / angle
    half = angle * 0.5
    q = torch.cat([torch.cos(half), torch.sin(half) * axis], dim=1)
    q = q / q.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    return torch.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
                        2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                        2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], dim=1).view(-1, 3, 3)


def _safe_div(num, den, eps=1.0e-6):
    """konia_transform.py:343-347 (note: the eps bump is applied twice, as in the reference)."""
    den = den.clone()
    den[den.abs() < eps] += eps
    den[den.abs() < eps] += eps
    return num / den


def rotation_matrix_to_quaternion(R: torch.Tensor, eps: float = 1.0e-6) -> torch.Tensor:
    """konia_transform.py:349-443, WXYZ order."""
    m = R.reshape(*R.shape[:-2], 9)
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.chunk(m, 9, dim=-1)
    trace = m00 + m11 + m22

    def pos():
        sq = torch.sqrt((trace + 1.0).clamp_min(eps)) * 2.0
        return torch.cat((0.25 * sq, _safe_div(m21 - m12, sq), _safe_div(m02 - m20, sq), _safe_div(m10 - m01, sq)), -1)

    def c1():
        sq = torch.sqrt((1.0 + m00 - m11 - m22).clamp_min(eps)) * 2.0
        return torch.cat((_safe_div(m21 - m12, sq), 0.25 * sq, _safe_div(m01 + m10, sq), _safe_div(m02 + m20, sq)), -1)

    def c2():
        sq = torch.sqrt((1.0 + m11 - m00 - m22).clamp_min(eps)) * 2.0
        return torch.cat((_safe_div(m02 - m20, sq), _safe_div(m01 + m10, sq), 0.25 * sq, _safe_div(m12 + m21, sq)), -1)

    def c3():
        sq = torch.sqrt((1.0 + m22 - m00 - m11).clamp_min(eps)) * 2.0
        return torch.cat((_safe_div(m10 - m01, sq), _safe_div(m02 + m20, sq), _safe_div(m12 + m21, sq), 0.25 * sq), -1)

    w2 = torch.where(m11 > m22, c2(), c3())
    w1 = torch.where((m00 > m11) & (m00 > m22), c1(), w2)
    return torch.where(trace > 0.0, pos(), w1)


def _safe_atan2(y, x, eps=1e-6):
    y = y.clone()
    y[(y.abs() < eps) & (x.abs() < eps)] += eps
    return torch.atan2(y, x)


def quaternion_to_angle_axis(q: torch.Tensor, eps: float = 1.0e-6) -> torch.Tensor:
    """konia_transform.py:560-630, WXYZ order."""
    cos_t, q1, q2, q3 = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    s2 = q1 * q1 + q2 * q2 + q3 * q3
    s = torch.sqrt(s2.clamp_min(eps))
    two_theta = 2.0 * torch.where(cos_t < 0.0, _safe_atan2(-s, -cos_t), _safe_atan2(s, cos_t))
    k = torch.where(s2 > 0.0, _safe_div(two_theta, s, eps), 2.0 * torch.ones_like(s))
    return torch.stack((q1 * k, q2 * k, q3 * k), dim=-1)


def rotation_matrix_to_angle_axis(R: torch.Tensor) -> torch.Tensor:
    """konia_transform.py:316-340."""
    return quaternion_to_angle_axis(rotation_matrix_to_quaternion(R))


def perspective_projection(points, translation, focal_length, camera_center=None):
    """geometry.py:78-116 with rotation = identity: K @ ((p + t) / z), first two rows."""
    B = points.shape[0]
    if camera_center is None:
        camera_center = torch.zeros(B, 2, dtype=points.dtype)
    p = points + translation.unsqueeze(1)
    p = p / p[:, :, -1:]
    u = focal_length[:, None, 0] * p[:, :, 0] + camera_center[:, None, 0] * p[:, :, 2]
    v = focal_length[:, None, 1] * p[:, :, 1] + camera_center[:, None, 1] * p[:, :, 2]
    return torch.stack((u, v), dim=-1)
