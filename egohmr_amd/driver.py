"""The sample / decode / evaluate block of the reference's stage-2 driver as a reusable harness.

What it stands in for (test_egohmr.py; argparse, the EgoBody dataloader, open3d rendering and logging are out of scope):
    :241-266   S sampling loops per batch through ``diffusion.val_losses`` (optionally collision-guided), ``model.eval_coll`` per
               sample, ``pred_smpl_params`` stacked to [B, S, ...]
    :268-318   ground-truth bodies (male / female SMPL by ``gender``), pelvis alignment
    :291-301   second SMPL decode of all B*S predicted bodies, camera-frame vertices / joints
    :374-505   G-MPJPE / MPJPE / PA-MPJPE / V2V with their visible / invisible splits, std / APD diversity, contact score
    :507-560   the running means the script prints (`error_dict`, `diversity_dict`)
    :672-695   ``results_seed_{seed}.pkl``
Everything stays on the device until ``summary()`` / ``results()``; the reference moves every metric to numpy per batch and runs
the Procrustes alignment as a CPU loop of numpy SVDs (utils/pose_utils.py).  The S samples of a batch share one conditioning pass
(``FusedSampler.prepare`` caches it per batch), the reference re-runs ResNet-50 and the PointNet in every denoising step.
"""
from __future__ import annotations

import numpy as np
import torch

from . import io as eio
from . import metrics as M
from .geometry import perspective_projection


class Stage2Driver:
    def __init__(self, model, diffusion, smpl_neutral, smpl_male, smpl_female, num_samples: int = 5, timestep_respacing: str = "",
                 with_coap_grad: bool = False, cond_grad_weight: float = 2.0, eval_coll_loss: bool = False,
                 eval_contact_score: bool = True, eval_with_vis_mask_pa: bool = False, fx_norm_coeff: float = 1500.0,
                 batch_samples: bool = True, two_stage: bool = False):
        if eval_with_vis_mask_pa:
            raise NotImplementedError("reconstruction_error_with_vis_mask (utils/pose_utils.py) is not on the default path (test_egohmr.py:76)")
        self.model, self.diffusion = model, diffusion
        self.smpl_neutral, self.smpl_male, self.smpl_female = smpl_neutral, smpl_male, smpl_female
        self.S, self.respacing = int(num_samples), timestep_respacing
        self.guided, self.w = bool(with_coap_grad), float(cond_grad_weight)
        self.eval_coll_loss, self.eval_contact = bool(eval_coll_loss), bool(eval_contact_score)
        self.fx_norm_coeff = fx_norm_coeff
        self.batch_samples = bool(batch_samples)   # the S samples of a batch as one fused loop over S*B bodies (FusedSampler.run_samples)
        # --two_stage (test_egohmr.py:24, default True there): the batch carries `stage1_transl_full` [B,3] - the stage-1 (ProHMR-scene)
        # camera-frame translation, results.pkl['pred_cam_full_list'] read by egohmr_amd.io.load_stage1_cam - and the sampler is
        # conditioned on it instead of the ground-truth translation (:243-245)
        self.two_stage = bool(two_stage)
        self._acc = {}
        self._lists = {k: [] for k in ("pred_betas", "pred_global_orient", "pred_body_pose", "gt_cam_full", "pred_cam_full", "coll", "contact")}
        self._vis_counts = dict(joint_vis=0, joint_invis=0, vertex_vis=0, vertex_invis=0)

    # ------------------------------------------------------------------ :241-266
    @torch.no_grad()
    def sample(self, batch, noise_stacks=None):
        """-> ({'betas' [B,S,10], 'global_orient' [B,S,1,3,3], 'body_pose' [B,S,23,3,3]}, coll_ratio [B,S] float64 numpy)."""
        B = batch["img"].shape[0]
        out = {"betas": [], "global_orient": [], "body_pose": []}
        coll = np.zeros((B, self.S))
        batched = None
        ddim = self.respacing[0:4] == "ddim"
        if self.S > 1 and self.batch_samples and self.diffusion._fused_ok(self.model, 0, None, None, False, 0.0) and not (ddim and self.guided):
            # the S loops of the reference (test_egohmr.py:251-266) as ONE loop over S*B bodies (FusedSampler.run_samples): same noise
            # draws in the same order, same per-body arithmetic
            self.model.validation_setup()
            dev = self.model.device
            stacks = noise_stacks if noise_stacks is not None else [self.diffusion._draw_stack([B, 144], dev, None) for _ in range(self.S)]
            batched = self.model.fused_sampler.run_samples(self.diffusion, batch, list(stacks[: self.S]), ddim=ddim, guided=self.guided,
                                                           cond_grad_weight=self.w)
        for n in range(self.S):
            if batched is not None:
                o = batched[n]["other_outputs"]
            else:
                o = self.diffusion.val_losses(model=self.model, batch=batch, shape=[B, 144], progress=False, clip_denoised=False, cur_epoch=0,
                                              timestep_respacing=self.respacing, cond_fn_with_grad=self.guided, cond_grad_weight=self.w,
                                              noise_stack=None if noise_stacks is None else noise_stacks[n])
            if self.eval_coll_loss:
                coll[:, n] = np.array(self.model.eval_coll(o))
            for k in out:
                out[k].append(o["pred_smpl_params"][k].unsqueeze(1))
        return {k: torch.cat(v, dim=1) for k, v in out.items()}, coll

    # ------------------------------------------------------------------ :291-301
    @torch.no_grad()
    def decode(self, pred, transl):
        B, S = pred["betas"].shape[:2]
        o = self.smpl_neutral(betas=pred["betas"].reshape(-1, 10), body_pose=pred["body_pose"].reshape(-1, 23, 3, 3),
                              global_orient=pred["global_orient"].reshape(-1, 1, 3, 3), pose2rot=False)
        verts = o.vertices.reshape(B, S, -1, 3)
        j24 = o.joints.reshape(B, S, -1, 3)[:, :, 0:24, :]
        pelvis = j24[:, :, [0], :].clone()
        t = transl.unsqueeze(1).unsqueeze(1)
        return dict(vertices=verts, joints=j24, pelvis=pelvis, joints_align=j24 - pelvis, vertices_align=verts - pelvis,
                    vertices_full=verts + t, joints_full=j24 + t)

    # ------------------------------------------------------------------ :306-318
    @torch.no_grad()
    def ground_truth(self, batch, gt_cam_full):
        sp = batch["smpl_params"]
        kw = dict(global_orient=sp["global_orient"], transl=gt_cam_full, body_pose=sp["body_pose"], betas=sp["betas"])
        male, female = self.smpl_male(**kw), self.smpl_female(**kw)
        fem = (batch["gender"] == 1).view(-1, 1, 1)
        joints = torch.where(fem, female.joints, male.joints)
        verts = torch.where(fem, female.vertices, male.vertices)
        j24 = joints[:, :24, :]
        pelvis = j24[:, [0], :].clone()
        return dict(joints=j24, vertices=verts, pelvis=pelvis, joints_align=j24 - pelvis, vertices_align=verts - pelvis)

    # ------------------------------------------------------------------ one batch: :230-505
    @torch.no_grad()
    def step(self, batch, noise_stacks=None):
        dev = batch["img"].device
        B = batch["img"].shape[0]
        gt_cam_full = batch["smpl_params"]["transl"].clone()                       # :238
        if self.two_stage:                                                         # :243-245: replace the gt camera translation by stage 1's
            if "stage1_transl_full" not in batch:
                raise KeyError("two_stage=True needs batch['stage1_transl_full'] [B,3] (egohmr_amd.io.load_stage1_cam reads it from the stage-1 results.pkl)")
            batch["smpl_params"]["transl"] = batch["stage1_transl_full"].to(dev).float()
            self._lists["pred_cam_full"].append(batch["smpl_params"]["transl"])    # :302-303
        pred, coll = self.sample(batch, noise_stacks)
        p = self.decode(pred, batch["smpl_params"]["transl"])
        g = self.ground_truth(batch, gt_cam_full)
        # visibility of the ground truth in the full image (:374-389)
        focal = (batch["fx"] * self.fx_norm_coeff).unsqueeze(-1).repeat(1, 2)
        center = torch.stack([batch["cam_cx"], batch["cam_cy"]], dim=-1)
        zero = torch.zeros(B, 3, device=dev)
        inside = lambda uv: (uv[..., 0] >= 0) & (uv[..., 0] < 1920) & (uv[..., 1] >= 0) & (uv[..., 1] < 1080)
        jvis = inside(perspective_projection(g["joints"], zero, focal, center))                  # [B,24]
        vvis = inside(perspective_projection(g["vertices"], zero, focal, center))                # [B,V]
        S = self.S
        # :399-449 as four launches (csrc/eval.hip): per-point error, its mean and its sums over the visible / invisible points; PA-MPJPE's similarity
        # transform (utils/pose_utils.py:10-66) in float64 inside the kernel - the reference copies to the host and loops numpy SVDs per sample
        res = {}
        for k, r in (("g_mpjpe", M.point_errors(p["joints_full"], g["joints"], points=24, mask=jvis)),                       # :399
                     ("mpjpe", M.point_errors(p["joints_align"], g["joints_align"], points=24, mask=jvis)),                  # :409
                     ("v2v", M.point_errors(p["vertices_align"], g["vertices_align"], mask=vvis)),                           # :441
                     ("pa_mpjpe", M.procrustes(p["joints_align"][:, :, :24].contiguous(), g["joints_align"][:, :24].contiguous(), mask=jvis))):   # :418-431
            res[k], res[k + "_vis_sum"], res[k + "_invis_sum"] = r["mean"], r["vis_sum"], r["invis_sum"]                     # [B,S] each
        if S > 1:                                                                               # :453-494: (std, apd) per joint selection, one launch each
            ja = p["joints_align"][:, :, :24].contiguous()
            res["std_joints"], res["apd_joints"] = M.diversity(ja)
            res["std_joints_vis"], res["apd_joints_vis"] = M.diversity(ja, jvis)
            res["std_joints_invis"], res["apd_joints_invis"] = M.diversity(ja, jvis, invert=True)
        else:
            res["std_joints"] = res["std_joints_vis"] = res["std_joints_invis"] = torch.full((B,), float("nan"), device=dev)
        if self.eval_contact:                                                                   # :496-505
            scene = batch["scene_pcd_verts_full"].unsqueeze(1).expand(-1, S, -1, -1).reshape(B * S, -1, 3)
            res["contact"] = M.contact_score(p["vertices_full"].reshape(B * S, -1, 3), scene).reshape(B, S).float()
        res["coll"] = torch.from_numpy(coll).to(dev)
        for k, v in res.items():
            self._acc.setdefault(k, []).append(v)
        self._vis_counts["joint_vis"] += int(jvis.sum())
        self._vis_counts["joint_invis"] += B * 24 - int(jvis.sum())
        self._vis_counts["vertex_vis"] += int(vvis.sum())
        self._vis_counts["vertex_invis"] += B * vvis.shape[1] - int(vvis.sum())
        self._lists["pred_betas"].append(pred["betas"])
        self._lists["pred_global_orient"].append(pred["global_orient"])
        self._lists["pred_body_pose"].append(pred["body_pose"])
        self._lists["gt_cam_full"].append(gt_cam_full)
        return dict(pred=pred, decoded=p, gt=g, joint_vis_mask=jvis, vertex_vis_mask=vvis, **res)

    # ------------------------------------------------------------------ :507-560, :660-670
    def summary(self) -> dict:
        a = {k: torch.cat(v, 0).double().cpu().numpy() for k, v in self._acc.items()}
        S, c = self.S, self._vis_counts
        out = {}
        for name, key, vis_n, invis_n in (("G-MPJPE", "g_mpjpe", c["joint_vis"], c["joint_invis"]), ("MPJPE", "mpjpe", c["joint_vis"], c["joint_invis"]),
                                          ("PA-MPJPE", "pa_mpjpe", c["joint_vis"], c["joint_invis"]), ("V2V", "v2v", c["vertex_vis"], c["vertex_invis"])):
            out[name] = 1000 * a[key].mean()
            out[name + "-vis"] = 1000 * a[key + "_vis_sum"].sum() / max(vis_n, 1) / S
            out[name + "-invis"] = 1000 * a[key + "_invis_sum"].sum() / max(invis_n, 1) / S
        for name, key in (("std-joints", "std_joints"), ("std-joints-vis", "std_joints_vis"), ("std-joints-invis", "std_joints_invis"),
                          ("apd-joints", "apd_joints"), ("apd-joints-vis", "apd_joints_vis"), ("apd-joints-invis", "apd_joints_invis")):
            if key in a:
                v = a[key]
                out[name] = 1000 * v[~np.isnan(v)].mean() if (~np.isnan(v)).any() else float("nan")
        if "contact" in a:
            out["contact"] = a["contact"].mean()
        out["coll"] = a["coll"].mean()
        return out

    def results(self) -> dict:
        cat = lambda k: torch.cat(self._lists[k], 0)
        contact = torch.cat(self._acc["contact"], 0) if "contact" in self._acc else torch.zeros_like(torch.cat(self._acc["coll"], 0))
        return eio.results_dict(cat("pred_betas"), cat("pred_global_orient"), cat("pred_body_pose"), torch.cat(self._acc["coll"], 0), contact,
                                cat("gt_cam_full"), pred_cam_full=cat("pred_cam_full") if self.two_stage else None)      # :685-693

    def save(self, save_root: str, model_id: str, seed: int) -> str:
        return eio.save_results(save_root, model_id, seed, self.results())
