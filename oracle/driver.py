"""Oracle: CPU restatement of the sample / decode / evaluate block of the reference's stage-2 driver, test_egohmr.py:241-266
(S sampling loops), :291-318 (second decode of the B*S bodies, ground-truth bodies), :374-505 (metrics), :672-695 (results dict).
Test infrastructure only.  `test_egohmr.py` itself cannot be imported (module-level argparse, cv2 / pytorch3d / pyrender / the
dataset), so this follows it line by line on the oracle model / sampler / SMPL and numpy, per-item loops included."""
from __future__ import annotations

import numpy as np
import torch

from oracle import geometry as geo
from oracle import metrics as om
from oracle import sampler as osamp


def run_batch(model, smpl_neutral, smpl_male, smpl_female, batch, tables, noise_stacks, respacing, num_samples, guided=False,
              cond_grad_weight=2.0, guide_reduction="mean", guide_all_points=False, eval_coll=False, fx_norm_coeff=1500.0, two_stage=False):
    B = batch["img"].shape[0]
    S = num_samples
    gt_cam_full = batch["smpl_params"]["transl"].clone()                                   # :238
    pred_cam_full = None
    if two_stage:                                                                            # :243-245
        batch["smpl_params"]["transl"] = batch["stage1_transl_full"]
        pred_cam_full = batch["stage1_transl_full"]
    outs = {"betas": [], "global_orient": [], "body_pose": []}
    coll = np.zeros((B, S))
    for n in range(S):                                                                       # :251-263
        o = osamp.val_losses(model, batch, tables, noise_stacks[n], respacing, cond_fn_with_grad=guided, cond_grad_weight=cond_grad_weight,
                             guide_reduction=guide_reduction, guide_all_points=guide_all_points)
        if eval_coll:
            coll[:, n] = np.array(model.eval_coll(o))
        for k in outs:
            outs[k].append(o["pred_smpl_params"][k].unsqueeze(1))
    pred = {k: torch.cat(v, dim=1) for k, v in outs.items()}                                 # :264-266
    po = smpl_neutral(betas=pred["betas"].reshape(-1, 10), body_pose=pred["body_pose"].reshape(-1, 23, 3, 3),
                      global_orient=pred["global_orient"].reshape(-1, 1, 3, 3))               # :291-293
    pv = po.vertices.reshape(B, S, -1, 3)
    pj = po.joints.reshape(B, S, -1, 3)[:, :, 0:24, :]
    ppel = pj[:, :, [0], :].clone()
    pja, pva = pj - ppel, pv - ppel
    t = batch["smpl_params"]["transl"].unsqueeze(1).unsqueeze(1)
    pvf, pjf = pv + t, pj + t                                                                # :300-301
    sp = batch["smpl_params"]
    Rg = geo.aa_to_rotmat(sp["global_orient"].reshape(-1, 3)).reshape(B, 1, 3, 3)            # smplx pose2rot=True (default) at :307,:310
    Rb = geo.aa_to_rotmat(sp["body_pose"].reshape(-1, 3)).reshape(B, 23, 3, 3)
    male = smpl_male(betas=sp["betas"], body_pose=Rb, global_orient=Rg, transl=gt_cam_full)
    female = smpl_female(betas=sp["betas"], body_pose=Rb, global_orient=Rg, transl=gt_cam_full)
    gj, gv = male.joints.clone(), male.vertices.clone()
    fem = batch["gender"] == 1
    gj[fem], gv[fem] = female.joints[fem], female.vertices[fem]                               # :313-314
    gj = gj[:, :24, :]
    gpel = gj[:, [0], :].clone()
    gja, gva = gj - gpel, gv - gpel
    focal = (batch["fx"] * fx_norm_coeff).unsqueeze(-1).repeat(1, 2)                          # :237,:375
    center = torch.cat([batch["cam_cx"].unsqueeze(-1), batch["cam_cy"].unsqueeze(-1)], dim=-1)
    zero = torch.zeros(B, 3)
    j2d = geo.perspective_projection(gj, zero, focal, center)
    v2d = geo.perspective_projection(gv, zero, focal, center)
    jvis = (j2d[:, :, 0] >= 0) * (j2d[:, :, 0] < 1920) * (j2d[:, :, 1] >= 0) * (j2d[:, :, 1] < 1080)   # :386-389
    vvis = (v2d[:, :, 0] >= 0) * (v2d[:, :, 0] < 1920) * (v2d[:, :, 1] >= 0) * (v2d[:, :, 1] < 1080)
    res = {}
    gm = torch.sqrt(((pjf - gj.unsqueeze(1)) ** 2).sum(-1))                                   # :399
    mp = torch.sqrt(((pja - gja.unsqueeze(1)) ** 2).sum(-1))                                  # :409
    vv = torch.sqrt(((pva - gva.unsqueeze(1)) ** 2).sum(-1))                                  # :441
    S1 = pja.reshape(-1, 24, 3).double().numpy()
    S2 = gja.unsqueeze(1).repeat(1, S, 1, 1).reshape(-1, 24, 3).double().numpy()
    hat = np.stack([om.procrustes(a, b) for a, b in zip(S1, S2)])                             # utils/pose_utils.py:109-126, avg_joint=False
    pa = torch.from_numpy(np.sqrt(((hat - S2) ** 2).sum(-1)).reshape(B, S, 24))
    for k, v, m in (("g_mpjpe", gm, jvis), ("mpjpe", mp, jvis), ("pa_mpjpe", pa, jvis), ("v2v", vv, vvis)):
        res[k] = v.mean(-1).double().numpy()
        res[k + "_vis_sum"] = (v * m.unsqueeze(1)).sum(-1).double().numpy()
        res[k + "_invis_sum"] = (v * (~m).unsqueeze(1)).sum(-1).double().numpy()
    if S > 1:
        res["std_joints"] = torch.std(pja, dim=1, unbiased=True).mean(-1).mean(-1).numpy()    # :453-455
        sv, si, av, ai = [], [], [], []
        a = pja.numpy()
        pair = np.linalg.norm(a[:, None] - a[:, :, None], axis=-1)
        res["apd_joints"] = pair.sum(axis=(-1, -2, -3)) / a.shape[-2] / S / (S - 1) / 2       # :471-476
        for k in range(B):                                                                   # :457-494
            for mask, so, ao in ((jvis[k], sv, av), (~jvis[k], si, ai)):
                tmp = pja[k][:, mask]
                so.append(torch.std(tmp, dim=0, unbiased=True).mean(-1).mean(-1).item() if tmp.shape[1] else float("nan"))
                tn = tmp.numpy()
                pw = np.linalg.norm(tn[None] - tn[:, None], axis=-1)
                ao.append(pw.sum() / tn.shape[-2] / S / (S - 1) / 2 if tn.shape[1] else float("nan"))
        res.update(std_joints_vis=np.array(sv), std_joints_invis=np.array(si), apd_joints_vis=np.array(av), apd_joints_invis=np.array(ai))
    scene = batch["scene_pcd_verts_full"].unsqueeze(1).repeat(1, S, 1, 1).reshape(B * S, -1, 3)
    d2, _ = om.nn_dist2(pvf.reshape(B * S, -1, 3).numpy(), scene.numpy())                     # :496-505 (squared distances, threshold 0.02)
    res["contact"] = (d2.min(-1) < 0.02).reshape(B, S).astype(np.float64)
    res["coll"] = coll
    return dict(pred=pred, joints_align=pja, vertices=pv, gt_joints=gj, gt_vertices=gv, joint_vis_mask=jvis, vertex_vis_mask=vvis,
                gt_cam_full=gt_cam_full, pred_cam_full=pred_cam_full, joints_full=pjf, **res)
