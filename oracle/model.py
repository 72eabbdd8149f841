"""Oracle: EgoHMR stage-2 model forward on the CPU (eager torch).  Test infrastructure only.

A functional restatement over a flat ``state_dict`` (reference parameter names) of

* ``EgoHMR.forward``                models/egohmr/egohmr.py:173-303
* ``EgoHMR.guide_coll`` plumbing    models/egohmr/egohmr.py:517-570 (collision term pluggable)
* ``ModulatedGCN`` / ``_GraphConv`` / ``_ResGraphConv``   models/egohmr/modulated_gcn/modulated_gcn.py:8-116
* ``ModulatedGraphConv.forward``    models/egohmr/modulated_gcn/modulated_gcn_conv.py:39-50
* ``TimestepEmbedder`` / ``InputProcess`` / ``FCHeadBeta`` / ``TranslEnc``   egohmr.py:629-690
* ``ResNet.forward`` (Bottleneck [3,4,6,3])   models/resnet.py:60-150
* ``ResnetPointnet.forward`` / ``ResnetBlockFC``   models/respointnet.py:33-97
* the SMPL-tree adjacency            egohmr.py:86-93

``faithful=True`` re-runs both encoders inside every call exactly like the reference does;
``faithful=False`` caches the step-invariant encoder outputs per batch (results identical, it is
the same arithmetic evaluated once) - SURVEY.md section 0 finding 2.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import geometry as geo
from .smpl import SMPLOracle

OPENPOSE_TO_SMPL = [8, 12, 9, 8, 13, 10, 8, 14, 11, 8, 14, 11, 0, 5, 2, 0, 5, 2, 6, 3, 7, 4, 7, 4]          # egohmr.py:111
OPENPOSE_TO_SMPL_LOOSE = [8, 13, 10, 8, 13, 10, 8, 14, 11, 8, 14, 11, 1, 5, 2, 0, 5, 2, 6, 3, 7, 4, 7, 4]    # egohmr.py:114
SMPL_EDGES = [(0, 1), (0, 2), (0, 3), (1, 4), (2, 5), (3, 6), (4, 7), (5, 8), (6, 9), (7, 10), (8, 11), (9, 12),
              (9, 13), (9, 14), (12, 15), (13, 16), (14, 17), (16, 18), (17, 19), (18, 20), (19, 21), (20, 22),
              (21, 23)]                                                                                  # other_utils.py:86-108


def smpl_adjacency(dtype=torch.float32) -> torch.Tensor:
    """egohmr.py:86-93: symmetric tree adjacency, row-normalised, then diagonal forced to 1."""
    a = np.zeros((24, 24), dtype=np.float32)
    for i, j in SMPL_EDGES:
        a[i, j] = 1.0
    a = a + a.T * (a.T > a) - a * (a.T > a)
    rs = a.sum(1)
    rinv = np.where(rs > 0, 1.0 / np.where(rs > 0, rs, 1), 0.0).astype(np.float32)
    a = rinv[:, None] * a
    a = torch.tensor(a, dtype=torch.float32)
    eye = torch.eye(24)
    return (a * (1 - eye) + eye).to(dtype)


def _bn(x, sd, p, eps=1e-5):
    shape = [1, -1] + [1] * (x.dim() - 2)
    return (x - sd[p + ".running_mean"].view(shape)) / torch.sqrt(sd[p + ".running_var"].view(shape) + eps) \
        * sd[p + ".weight"].view(shape) + sd[p + ".bias"].view(shape)


# ------------------------------------------------------------------------------------------- encoders

def resnet50(sd, x, p="backbone."):
    """models/resnet.py:139-150 with Bottleneck :76-95 (eval-mode batch norm)."""
    x = F.conv2d(x, sd[p + "conv1.weight"], stride=2, padding=3)
    x = F.relu(_bn(x, sd, p + "bn1"))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for li, (blocks, stride) in enumerate([(3, 1), (4, 2), (6, 2), (3, 2)], 1):
        for b in range(blocks):
            q = f"{p}layer{li}.{b}."
            s = stride if b == 0 else 1
            out = F.relu(_bn(F.conv2d(x, sd[q + "conv1.weight"]), sd, q + "bn1"))
            out = F.relu(_bn(F.conv2d(out, sd[q + "conv2.weight"], stride=s, padding=1), sd, q + "bn2"))
            out = _bn(F.conv2d(out, sd[q + "conv3.weight"]), sd, q + "bn3")
            res = x
            if (q + "downsample.0.weight") in sd:
                res = _bn(F.conv2d(x, sd[q + "downsample.0.weight"], stride=s), sd, q + "downsample.1")
            x = F.relu(out + res)
    return x.mean(dim=(2, 3))


def _resblock_fc(sd, p, x):
    """respointnet.py:89-97."""
    net = F.linear(F.relu(x), sd[p + "fc_0.weight"], sd[p + "fc_0.bias"])
    dx = F.linear(F.relu(net), sd[p + "fc_1.weight"], sd[p + "fc_1.bias"])
    return F.linear(x, sd[p + "shortcut.weight"]) + dx


def resnet_pointnet(sd, pts, p="scene_enc."):
    """respointnet.py:33-59."""
    net = F.linear(pts, sd[p + "fc_pos_0.weight"], sd[p + "fc_pos_0.bias"])
    net = _resblock_fc(sd, p + "block_0.", net)
    for b in (1, 2, 3):
        pooled = net.max(dim=1, keepdim=True)[0].expand(net.size())
        net = _resblock_fc(sd, p + f"block_{b}.", torch.cat([net, pooled], dim=2))
    net = net.max(dim=1)[0]
    return F.linear(F.relu(net), sd[p + "fc_c.weight"], sd[p + "fc_c.bias"])


# ------------------------------------------------------------------------------------------- denoiser

def modulated_graph_conv(sd, p, x, adj):
    """modulated_gcn_conv.py:39-50."""
    W, M = sd[p + ".W"], sd[p + ".M"]
    h0 = torch.matmul(x, W[0])
    h1 = torch.matmul(x, W[1])
    a = adj + sd[p + ".adj2"]
    a = (a.T + a) / 2
    E = torch.eye(a.size(0), dtype=a.dtype)
    out = torch.matmul(a * E, M * h0) + torch.matmul(a * (1 - E), M * h1)
    return out + sd[p + ".bias"].view(1, 1, -1)


def _graph_conv(sd, p, x, adj):
    """modulated_gcn.py:21-28 (BatchNorm1d over channels in eval mode, ReLU; dropout p=0)."""
    y = modulated_graph_conv(sd, p + ".gconv", x, adj).transpose(1, 2)
    return F.relu(_bn(y, sd, p + ".bn").transpose(1, 2))


def non_local_block(sd, p, x):
    """NONLocalBlock2D(hid, sub_sample=False, bn_layer=True) applied to the joint axis (modulated_gcn.py:105-110 wraps [B,24,C]
    into [B,C,1,24]; nets/non_local_embedded_gaussian.py:60-85): embedded-Gaussian attention over the 24 joints,
    z = BN(W (softmax(theta phi^T) g)) + x.  x [B,24,C]; the 1x1 convolutions are per-joint linear maps."""
    lin = lambda n, v: F.linear(v, sd[p + n + ".weight"].flatten(1), sd[p + n + ".bias"])
    g_x, th, ph = lin("g", x), lin("theta", x), lin("phi", x)                     # [B,24,C/2]   :68-74
    f = torch.matmul(th, ph.transpose(1, 2))                                       # [B,24,24]   :75
    y = torch.matmul(F.softmax(f, dim=-1), g_x)                                    # :76-78
    w_y = lin("W.0", y)                                                            # :81 conv
    scale = sd[p + "W.1.weight"] / torch.sqrt(sd[p + "W.1.running_var"] + 1e-5)   # BatchNorm2d, eval
    w_y = (w_y - sd[p + "W.1.running_mean"]) * scale + sd[p + "W.1.bias"]
    return w_y + x                                                                 # :82


def modulated_gcn(sd, x, adj, p="diffusion_model.", num_blocks=4, nonlocal_layer=False):
    """modulated_gcn.py:99-116; nonlocal_layer=False in every shipped config."""
    out = _graph_conv(sd, p + "gconv_input.0", x, adj)
    for b in range(num_blocks):
        res = out
        out = _graph_conv(sd, f"{p}gconv_layers.{b}.gconv1", out, adj)
        out = _graph_conv(sd, f"{p}gconv_layers.{b}.gconv2", out, adj)
        out = res + out
    if nonlocal_layer:
        out = non_local_block(sd, p + "non_local.", out)                            # :104-110
    return modulated_graph_conv(sd, p + "gconv_output", out, adj)


def timestep_embedding(sd, t):
    """egohmr.py:642-643 (result [B,512] after the permute/squeeze at :178)."""
    e = sd["embed_timestep.sequence_pos_encoder.pe"][t][:, 0]
    e = F.linear(e, sd["embed_timestep.time_embed.0.weight"], sd["embed_timestep.time_embed.0.bias"])
    return F.linear(F.silu(e), sd["embed_timestep.time_embed.2.weight"], sd["embed_timestep.time_embed.2.bias"])


# ------------------------------------------------------------------------------------------- model

class EgoHMROracle:
    """Eval-mode EgoHMR with the test-time flags of test_egohmr.py:112-118
    (with_focal_length / with_bbox_info / with_cam_center = True, scene_cano=True,
    only_mask_img_cond=True, cond_mask_prob=0)."""

    FX_NORM_COEFF = 1500.0  # configs/prohmr.yaml:56

    def __init__(self, state_dict: dict, smpl_asset: dict, body_rep_mean, body_rep_std,
                 diffuse_fuse=True, pelvis_vis_loosen=True, dtype=torch.float32, faithful=True,
                 num_blocks=4, collision_loss=None, gcn_nonlocal_layer=False, with_bbox_info=True, with_cam_center=True,
                 only_mask_img_cond=True):
        self.dtype = dtype
        self.sd = {k: (torch.as_tensor(v).to(dtype) if np.asarray(v).dtype.kind == "f" else torch.as_tensor(v))
                   for k, v in state_dict.items()}
        self.smpl = SMPLOracle(smpl_asset, dtype)
        self.mean = torch.as_tensor(body_rep_mean).to(dtype)
        self.std = torch.as_tensor(body_rep_std).to(dtype)
        self.diffuse_fuse = diffuse_fuse
        self.with_bbox_info, self.with_cam_center, self.only_mask_img_cond = with_bbox_info, with_cam_center, only_mask_img_cond   # egohmr.py:31,36
        self.op2smpl = OPENPOSE_TO_SMPL_LOOSE if pelvis_vis_loosen else OPENPOSE_TO_SMPL
        self.adj = smpl_adjacency(dtype)
        self.faithful = faithful
        self.num_blocks = num_blocks
        self.nonlocal_layer = gcn_nonlocal_layer                                      # egohmr.py:37,:99
        self.collision_loss = collision_loss
        self._cache_key = None
        self._cache = None

    def parameters(self):  # device discovery, gaussian_diffusion.py:472-473
        return iter([self.sd["input_process.poseEmbedding.weight"]])

    def validation_setup(self):
        pass

    def _encode(self, batch):
        key = id(batch)
        if not self.faithful and self._cache_key == key:
            return self._cache
        dt, sd = self.dtype, self.sd
        img_feats = resnet50(sd, batch["img"].to(dt))                                  # egohmr.py:183
        transl = batch["smpl_params"]["transl"].to(dt)
        scene = batch["scene_pcd_verts_full"].to(dt) - transl.unsqueeze(1)             # :211 (scene_cano)
        scene_feats = resnet_pointnet(sd, scene)                                       # :214
        h = F.relu(F.linear(transl, sd["transl_enc.layers.0.weight"], sd["transl_enc.layers.0.bias"]))
        transl_feat = F.linear(h, sd["transl_enc.layers.2.weight"], sd["transl_enc.layers.2.bias"])  # :217
        fx = batch["fx"].to(dt)
        ofx = fx * self.FX_NORM_COEFF
        cam = [fx.unsqueeze(1)]                                                        # :195-205: each part is PREpended
        if self.with_bbox_info:
            cam = [torch.stack([batch["box_center"][:, 0].to(dt) / ofx, batch["box_center"][:, 1].to(dt) / ofx,
                                batch["box_size"].to(dt) / ofx], -1)] + cam
        if self.with_cam_center:
            cam = [torch.stack([batch["cam_cx"].to(dt) / ofx, batch["cam_cy"].to(dt) / ofx], -1)] + cam
        cam = torch.cat(cam, dim=1)                                                    # -> [B, 1 (+3) (+2)]
        out = dict(img_feats=img_feats, scene=scene, scene_feats=scene_feats, transl_feat=transl_feat, cam=cam,
                   transl=transl)
        self._cache_key, self._cache = key, out
        return out

    def visibility(self, batch):
        vis = batch["orig_keypoints_2d"][:, :, -1] > 0                                  # :186
        vis = vis.clone()
        vis[:, 8] = True                                                               # :187
        return vis[:, self.op2smpl]                                                    # :188

    def __call__(self, batch, timesteps):
        return self.forward(batch, timesteps)

    def forward(self, batch, timesteps):
        sd, dt = self.sd, self.dtype
        B = batch["img"].shape[0]
        temb = timestep_embedding(sd, timesteps).unsqueeze(1).repeat(1, 24, 1)          # :178-179
        enc = self._encode(batch)
        vis = self.visibility(batch)
        batch["vis_mask_smpl"] = vis                                                   # :189
        img24 = enc["img_feats"].unsqueeze(1).repeat(1, 24, 1) * vis.unsqueeze(-1).to(dt)   # :190-191
        other = torch.cat([enc["scene_feats"], enc["transl_feat"], enc["cam"]], dim=1)  # :220-221
        cond = torch.cat([img24, other.unsqueeze(1).repeat(1, 24, 1)], dim=-1)          # :222-223  [B,24,2694]
        x_t = batch["x_t"].to(dt).reshape(B, 24, -1)
        x_feat = F.linear(x_t, sd["input_process.poseEmbedding.weight"], sd["input_process.poseEmbedding.bias"])
        out = modulated_gcn(sd, torch.cat([cond, x_feat, temb], dim=-1), self.adj, num_blocks=self.num_blocks, nonlocal_layer=self.nonlocal_layer)  # :236-237
        if self.diffuse_fuse:                                                          # :239-254
            cond_u = cond.clone()
            if self.only_mask_img_cond:
                cond_u[:, :, 0:2048] = 0                                               # mask_cond(force_mask=True), :151-155
            else:
                cond_u = torch.zeros_like(cond)                                        # :156-157
            out_u = modulated_gcn(sd, torch.cat([cond_u, x_feat, temb], dim=-1), self.adj, num_blocks=self.num_blocks, nonlocal_layer=self.nonlocal_layer)
            out_c = out
            out = out_u + 0 * (out_c - out_u)                                          # guidance_param = 0
            m = vis.unsqueeze(-1).repeat(1, 1, 6).reshape(B, -1)
            out = out.reshape(B, -1)
            out[m] = out_c.reshape(B, -1)[m]
        x0 = out.reshape(B, -1)
        pose6d = x0 * self.std + self.mean                                             # :258
        R = geo.rot6d_to_rotmat(pose6d, "diffusion").view(B, 24, 3, 3)                 # :260
        feats_beta = torch.cat([enc["img_feats"], other], dim=1)                       # :263-264
        h = F.relu(F.linear(feats_beta, sd["beta_layer.layers.0.weight"], sd["beta_layer.layers.0.bias"]))
        betas = F.linear(h, sd["beta_layer.layers.2.weight"], sd["beta_layer.layers.2.bias"]) + sd["beta_layer.init_betas"]
        so = self.smpl(betas=betas, body_pose=R[:, 1:], global_orient=R[:, [0]], return_full_pose=True)   # :276
        self.scene_pcd_verts = enc["scene"]
        focal = (batch["fx"].to(dt).unsqueeze(-1).repeat(1, 2)) * self.FX_NORM_COEFF   # :283-285
        center = torch.stack([batch["cam_cx"].to(dt), batch["cam_cy"].to(dt)], dim=-1)
        kp2d = geo.perspective_projection(so.joints, enc["transl"], focal, center)      # :295-298
        kp2d = torch.stack([kp2d[:, :, 0] / 1920 - 0.5, kp2d[:, :, 1] / 1080 - 0.5], dim=-1)
        return {
            "pred_x_start": x0,
            "pred_smpl_params": {"global_orient": R[:, [0]].clone(), "body_pose": R[:, 1:].clone(), "betas": betas.clone()},
            "pred_pose_6d": pose6d,
            "pred_keypoints_3d": so.joints,
            "pred_vertices": so.vertices,
            "pred_keypoints_3d_full": so.joints + enc["transl"].unsqueeze(1),
            "pred_keypoints_2d_full": kp2d,
        }

    # egohmr.py:517-570.  ``collision_loss(points[1,n,3], verts[1,V,3], joints, full_pose_aa) -> scalar``
    # stands in for ``smpl.coap.collision_loss`` (learned network, unavailable offline).
    GRAD_ZERO_JOINTS = [0, 3, 6, 9, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23]

    def guide_coll(self, batch, output, t, compute_grad="x_t", reduction="mean", all_points=False):
        """reduction='mean', all_points=False: egohmr.py:517-570 (COAP).  reduction='sum', all_points=True: the VolSMPL twin
        (egohmr_volsmpl.py:582-629): one batched loss over ALL scene points, `-loss.sum()`."""
        assert self.collision_loss is not None
        with torch.enable_grad():
            x = (batch["x_t"] if compute_grad == "x_t" else output["pred_x_start"]).detach().to(self.dtype)
            B = x.shape[0]
            # Quirk kept from egohmr.py:523-528,562: ``x_t`` is re-bound to ``x_t*std+mean`` before
            # autograd.grad, so the gradient is w.r.t. the DE-NORMALISED 6-D pose (no factor std).
            x = (x * self.std + self.mean).requires_grad_()
            R = geo.rot6d_to_rotmat(x, "diffusion").view(B, 24, 3, 3)
            so = self.smpl(betas=output["pred_smpl_params"]["betas"].detach(), body_pose=R[:, 1:],
                           global_orient=R[:, [0]], return_full_pose=True)
            aa = geo.rotation_matrix_to_angle_axis(so.full_pose.reshape(-1, 3, 3)).reshape(B, -1)
            losses = []
            for i in range(B):
                v = so.vertices[[i]]
                bb_min = v.min(1).values.reshape(1, 3).detach()
                bb_max = v.max(1).values.reshape(1, 3).detach()
                pts = self.scene_pcd_verts[[i]]
                inds = (pts >= bb_min).all(-1) & (pts <= bb_max).all(-1)
                if all_points:                                               # egohmr_volsmpl.py:609-612: no selection
                    inds = torch.ones_like(inds)
                if inds.any():
                    losses.append(self.collision_loss(pts[inds].unsqueeze(0), v, so.joints[[i]], aa[[i]]))
                else:
                    losses.append(torch.zeros((), dtype=self.dtype))
            loss = torch.stack(losses)
            if int((loss == 0).sum()) < B:
                red = loss.mean() if reduction == "mean" else loss.sum()     # egohmr.py:562 / egohmr_volsmpl.py:618
                g = torch.autograd.grad([-red], [x])[0].reshape(-1, 24, 6).clone()
                g[:, 3:] = g[:, 3:] * 2                                      # :564-565 (joint slices)
                g[:, self.GRAD_ZERO_JOINTS] = 0                              # :567
                return g.reshape(-1, 144), loss.detach()
            return torch.zeros(B, 144, dtype=self.dtype), loss.detach()

    def eval_coll(self, output, tau=0.05):
        """egohmr.py:487-514 (and eval_coll_volsmpl, egohmr_volsmpl.py:548-579) with the proxy: per item, bbox-selected scene points
        closer than tau to the body, over N."""
        from oracle.collision import proxy_min_dist
        p = output["pred_smpl_params"]
        so = self.smpl(betas=p["betas"], body_pose=p["body_pose"], global_orient=p["global_orient"])
        out = []
        for i in range(so.vertices.shape[0]):
            v = so.vertices[i]
            pts = self.scene_pcd_verts[i]
            inds = (pts >= v.min(0).values).all(-1) & (pts <= v.max(0).values).all(-1)
            out.append(float((proxy_min_dist(pts[inds], v) < tau).sum()) / pts.shape[0] if inds.any() else 0.0)
        return out
