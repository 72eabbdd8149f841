"""Oracle: SMPL linear blend skinning on the CPU (torch, any float dtype).  Test infrastructure only.

PARITY UNPINNED BY THE REFERENCE: the arithmetic is third-party - pip ``smplx==0.1.28``
(environment.yml:197), not vendored under /root/reference and not installed in this image.  This is
a restatement of the published smplx algorithm, with the reference's call sites as the anchor
(models/egohmr/egohmr.py:105-107 create, :276 / :492 / :537 call with rotation matrices,
``pose2rot=False``, ``return_full_pose=True``; test_egohmr.py:291 decode):

    smplx/lbs.py              blend_shapes, vertices2joints, batch_rigid_transform, lbs
    smplx/body_models.py      SMPL.forward (full_pose = cat[global_orient, body_pose]; default zero
                              ``transl`` parameter added to joints and vertices)
    smplx/vertex_joint_selector.py  21 extra joints picked from vertices -> 45 joints
"""
from __future__ import annotations

from types import SimpleNamespace

import torch


class SMPLOracle:
    def __init__(self, asset: dict, dtype=torch.float32):
        t = lambda k: torch.as_tensor(asset[k]).to(dtype)
        self.dtype = dtype
        self.v_template = t("v_template")                      # [V,3]
        self.shapedirs = t("shapedirs")                        # [V,3,10]
        self.posedirs = t("posedirs")                          # [207, V*3]
        self.J_regressor = t("J_regressor")                    # [24,V]
        self.lbs_weights = t("lbs_weights")                    # [V,24]
        self.parents = [int(p) for p in asset["parents"]]
        self.extra_joints_idxs = torch.as_tensor(asset["extra_joints_idxs"]).long()
        self.faces = asset.get("faces")

    def __call__(self, betas, body_pose, global_orient, transl=None, return_full_pose=False, pose2rot=False):
        assert not pose2rot, "the hot path always passes rotation matrices (egohmr.py:276)"
        B = betas.shape[0]
        dt = self.dtype
        betas = betas.to(dt)
        rot = torch.cat([global_orient.reshape(B, 1, 3, 3), body_pose.reshape(B, 23, 3, 3)], dim=1).to(dt)

        # lbs(): shape blend shapes and joint regression
        v_shaped = self.v_template[None] + torch.einsum("bl,mkl->bmk", betas, self.shapedirs)
        J = torch.einsum("bik,ji->bjk", v_shaped, self.J_regressor)                    # [B,24,3]
        # pose-corrective blend shapes
        ident = torch.eye(3, dtype=dt)
        pose_feature = (rot[:, 1:] - ident).reshape(B, -1)                             # [B,207]
        v_posed = v_shaped + torch.matmul(pose_feature, self.posedirs).view(B, -1, 3)
        # batch_rigid_transform(): kinematic chain
        rel = J.clone()
        rel[:, 1:] = rel[:, 1:] - J[:, self.parents[1:]]
        T = torch.zeros(B, 24, 4, 4, dtype=dt)
        T[:, :, :3, :3] = rot
        T[:, :, :3, 3] = rel
        T[:, :, 3, 3] = 1.0
        chain = [T[:, 0]]
        for i in range(1, 24):
            chain.append(torch.matmul(chain[self.parents[i]], T[:, i]))
        G = torch.stack(chain, dim=1)                                                   # [B,24,4,4]
        posed_joints = G[:, :, :3, 3]
        J_h = torch.cat([J, torch.zeros(B, 24, 1, dtype=dt)], dim=2).unsqueeze(-1)      # [B,24,4,1]
        A = G.clone()
        A[:, :, :, 3:4] = A[:, :, :, 3:4] - torch.matmul(G, J_h)                        # rel_transforms
        # skinning
        Tv = torch.matmul(self.lbs_weights[None].expand(B, -1, -1), A.view(B, 24, 16)).view(B, -1, 4, 4)
        v_h = torch.cat([v_posed, torch.ones(B, v_posed.shape[1], 1, dtype=dt)], dim=2)
        verts = torch.matmul(Tv, v_h.unsqueeze(-1))[:, :, :3, 0]
        # VertexJointSelector + default zero transl
        joints = torch.cat([posed_joints, verts[:, self.extra_joints_idxs]], dim=1)     # [B,45,3]
        if transl is not None:
            joints = joints + transl.to(dt).unsqueeze(1)
            verts = verts + transl.to(dt).unsqueeze(1)
        return SimpleNamespace(vertices=verts, joints=joints, betas=betas,
                               global_orient=global_orient, body_pose=body_pose,
                               full_pose=rot if return_full_pose else None,
                               A=A[:, :, :3, :], v_shaped=v_shaped, J=J)
