"""Rotation-representation helpers with the reference's names (utils/geometry.py).

``rot6d_to_rotmat`` runs the gfx950 kernel (csrc/smpl.hip: rot6d_kernel) - it is on the hot path
(utils/geometry.py:47-66, called at models/egohmr/egohmr.py:260 and :529); ``rotation_matrix_to_angle_axis``
(utils/konia_transform.py:316-340, the collision models' ``full_pose`` feed) runs csrc/eval.hip's kernel with a
hand-written VJP.  The remaining helpers (a slice, the training-side aa_to_rotmat, the output garnish
perspective_projection outside the fused packer) are thin torch expressions kept for API completeness.
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


class _RotmatToAngleAxis(torch.autograd.Function):
    @staticmethod
    def forward(ctx, R):
        R = _lib.f32(R).reshape(-1, 3, 3)
        n = R.shape[0]
        aa = torch.empty(n, 3, device=R.device, dtype=torch.float32)
        with _lib.on_device(R.device):
            _lib.check(_lib.lib().ehm_rotmat_to_angle_axis(_lib.ptr(R), _lib.ptr(aa), n, _lib.stream_ptr()), "ehm_rotmat_to_angle_axis")
        ctx.save_for_backward(R)
        return aa

    @staticmethod
    def backward(ctx, gaa):
        (R,) = ctx.saved_tensors
        gaa = _lib.f32(gaa)
        gR = torch.empty_like(R)
        with _lib.on_device(R.device):
            _lib.check(_lib.lib().ehm_rotmat_to_angle_axis_bwd(_lib.ptr(R), _lib.ptr(gaa), _lib.ptr(gR), R.shape[0], _lib.stream_ptr()), "ehm_rotmat_to_angle_axis_bwd")
        return gR


def rotation_matrix_to_angle_axis(rotation_matrix: torch.Tensor) -> torch.Tensor:
    """utils/konia_transform.py:316-340 (-> rotation_matrix_to_quaternion :349-443, quaternion_to_angle_axis :560-630) on the HIP kernel
    ehm_rotmat_to_angle_axis (csrc/eval.hip), autograd-capable (ehm_rotmat_to_angle_axis_bwd: the VJP through the branch the forward took).
    Feeds ``full_pose`` of COAP-style collision models (egohmr.py:495, :540) under torch.enable_grad(); (*, 3, 3) -> (*, 3)."""
    if rotation_matrix.shape[-2:] != (3, 3):
        raise ValueError(f"Input size must be a (*, 3, 3) tensor. Got {rotation_matrix.shape}")
    if not rotation_matrix.is_cuda:
        raise _lib.EgoHMRHipError("rotation_matrix_to_angle_axis runs on the HIP kernel (csrc/eval.hip); the CPU restatement is oracle/geometry.py")
    return _RotmatToAngleAxis.apply(rotation_matrix).reshape(*rotation_matrix.shape[:-2], 3)
