"""EgoHMR stage-2 model with the reference's constructor / call surface, executing on libegohmr_hip.

Reference seams honoured (models/egohmr/egohmr.py):
  EgoHMR(cfg, device, body_rep_mean, body_rep_std, with_focal_length, with_bbox_info, with_cam_center, ...)   :29-40
  .forward(batch, timesteps) -> dict(pred_x_start, pred_smpl_params, pred_pose_6d, pred_keypoints_3d,
                                     pred_vertices, pred_keypoints_3d_full, pred_keypoints_2d_full)          :173-303
  .validation_setup()  :475-484     .guide_coll(batch, output, t, compute_grad)  :517-605
  .eval_coll(output)   :487-514     .parameters() / load_state_dict(strict=False) with the reference's names
plus ``fused_sampler`` (build extension) which diffusion.py uses to run the whole sampling loop natively.

What is different by design: the encoders (ResNet-50, scene PointNet), the translation / beta heads and
the conditioning + timestep slices of the GCN input conv do not depend on x_t, so they are evaluated once
per batch (``prepare``) instead of once per denoising step (egohmr.py:183,:214 sit inside forward); the
denoiser, rot6d->rotmat, SMPL LBS and the sampler update are hand-written HIP kernels (csrc/*.hip).
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, geometry, synthetic
from . import smpl as smpl_mod
from .encoders import ResnetPointnet, ResNet50Features

OPENPOSE_TO_SMPL = [8, 12, 9, 8, 13, 10, 8, 14, 11, 8, 14, 11, 0, 5, 2, 0, 5, 2, 6, 3, 7, 4, 7, 4]           # egohmr.py:111
OPENPOSE_TO_SMPL_LOOSE = [8, 13, 10, 8, 13, 10, 8, 14, 11, 8, 14, 11, 1, 5, 2, 0, 5, 2, 6, 3, 7, 4, 7, 4]     # egohmr.py:114
IMG_DIM = 2048          # columns of the GCN input feature: img 2048 | scene + transl + cam | x_t embed 512 | timestep embed 512 (EgoHMR.cond_split)
PRECISIONS = {"f32": 0, "f16x3": 1, "f16": 2}
GRAD_ZERO_JOINTS = [0, 3, 6, 9, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23]                               # egohmr.py:567


def default_cfg():
    """The yacs keys the sampling path reads (configs/prohmr.yaml:41-56)."""
    return SimpleNamespace(MODEL=SimpleNamespace(BACKBONE=SimpleNamespace(NUM_LAYERS=50, OUT_CHANNELS=2048)),
                           CAM=SimpleNamespace(FX_NORM_COEFF=1500.0), EXTRA=SimpleNamespace(FOCAL_LENGTH=5000.0),
                           TRAIN=SimpleNamespace(LR=1e-4, WEIGHT_DECAY=1e-4))


def smpl_tree_adjacency() -> torch.Tensor:
    """egohmr.py:86-93: symmetric SMPL-tree adjacency, rows normalised, diagonal forced to one."""
    a = np.zeros((24, 24), dtype=np.float32)
    for p, c in synthetic.SMPL_EDGES:
        a[p, c] = a[c, p] = 1.0
    a = a / a.sum(1, keepdims=True)
    np.fill_diagonal(a, 1.0)
    return torch.from_numpy(a)


# ---------------------------------------------------------------------------------------------- parameter holders
class ModulatedGraphConv(nn.Module):
    """Parameters of modulated_gcn_conv.py:16-37 (the arithmetic lives in csrc/gcn.hip)."""

    def __init__(self, in_features, out_features, adj):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.W = nn.Parameter(torch.empty(2, in_features, out_features))
        self.M = nn.Parameter(torch.empty(adj.size(0), out_features))
        self.adj2 = nn.Parameter(torch.full_like(adj, 1e-6))
        self.bias = nn.Parameter(torch.empty(out_features))
        nn.init.xavier_uniform_(self.W.data, gain=1.414)
        nn.init.xavier_uniform_(self.M.data, gain=1.414)
        bound = 1.0 / np.sqrt(out_features)
        nn.init.uniform_(self.bias, -bound, bound)


class _GraphConv(nn.Module):
    def __init__(self, adj, cin, cout):
        super().__init__()
        self.gconv = ModulatedGraphConv(cin, cout, adj)
        self.bn = nn.BatchNorm1d(cout)


class _ResGraphConv(nn.Module):
    def __init__(self, adj, dim):
        super().__init__()
        self.gconv1 = _GraphConv(adj, dim, dim)
        self.gconv2 = _GraphConv(adj, dim, dim)


class _NonLocalBlock(nn.Module):
    """Parameter tree of NONLocalBlock2D(in_channels=hid, sub_sample=False, bn_layer=True)
    (nets/non_local_embedded_gaussian.py:6-58): g / theta / phi 1x1 convs hid -> hid/2, W = Sequential(conv hid/2 -> hid, BatchNorm2d)
    with the reference's zero-initialised BatchNorm affine (identity block until trained)."""

    def __init__(self, hid_dim):
        super().__init__()
        ci = max(hid_dim // 2, 1)
        self.inter_channels = ci
        self.g = nn.Conv2d(hid_dim, ci, 1)
        self.theta = nn.Conv2d(hid_dim, ci, 1)
        self.phi = nn.Conv2d(hid_dim, ci, 1)
        self.W = nn.Sequential(nn.Conv2d(ci, hid_dim, 1), nn.BatchNorm2d(hid_dim))
        nn.init.constant_(self.W[1].weight, 0)
        nn.init.constant_(self.W[1].bias, 0)


class ModulatedGCN(nn.Module):
    """modulated_gcn.py:60-97 parameter tree: gconv_input.0, gconv_layers.{b}.gconv{1,2}, gconv_output."""

    def __init__(self, adj, in_dim, out_dim=6, hid_dim=1024, num_layers=4, nonlocal_layer=False, p_dropout=0.0):
        super().__init__()
        self.register_buffer("adj", adj.clone(), persistent=False)
        self.in_dim, self.hid_dim, self.out_dim, self.num_layers = in_dim, hid_dim, out_dim, num_layers
        self.gconv_input = nn.Sequential(_GraphConv(adj, in_dim, hid_dim))
        self.gconv_layers = nn.Sequential(*[_ResGraphConv(adj, hid_dim) for _ in range(num_layers)])
        self.gconv_output = ModulatedGraphConv(hid_dim, out_dim, adj)
        self.nonlocal_layer = bool(nonlocal_layer)
        if self.nonlocal_layer:                                     # modulated_gcn.py:93-94, reference parameter names
            self.non_local = _NonLocalBlock(hid_dim)

    def forward(self, x):
        raise NotImplementedError("the denoiser runs through EgoHMR.forward / EgoHMR.fused_sampler (hoisted input conv)")


class PositionalEncoding(nn.Module):
    def __init__(self, d_model, dropout=0.1, max_len=5000):
        super().__init__()
        self.register_buffer("pe", torch.from_numpy(synthetic.positional_table(max_len, d_model)))


class TimestepEmbedder(nn.Module):
    def __init__(self, latent_dim, sequence_pos_encoder):
        super().__init__()
        self.sequence_pos_encoder = sequence_pos_encoder
        self.time_embed = nn.Sequential(nn.Linear(latent_dim, latent_dim), nn.SiLU(), nn.Linear(latent_dim, latent_dim))

    def forward(self, timesteps):
        return self.time_embed(self.sequence_pos_encoder.pe[timesteps]).permute(1, 0, 2)   # egohmr.py:642-643


class InputProcess(nn.Module):
    def __init__(self, input_dim, latent_dim):
        super().__init__()
        self.poseEmbedding = nn.Linear(input_dim, latent_dim)


class FCHeadBeta(nn.Module):
    def __init__(self, in_dim):
        super().__init__()
        self.layers = nn.Sequential(nn.Linear(in_dim, 1024), nn.ReLU(), nn.Linear(1024, 10))
        self.register_buffer("init_betas", torch.zeros(1, 10))   # data/smpl_mean_params.npz['shape'] in the reference (:669-671)

    def forward(self, feats, pred_pose=None):
        return self.layers(feats) + self.init_betas


class TranslEnc(nn.Module):
    def __init__(self, in_dim=3, out_dim=128):
        super().__init__()
        self.layers = nn.Sequential(nn.Linear(in_dim, 64), nn.ReLU(), nn.Linear(64, out_dim))

    def forward(self, x):
        return self.layers(x)


# ---------------------------------------------------------------------------------------------- model
class EgoHMR(nn.Module):
    def __init__(self, cfg=None, device=None, body_rep_mean=None, body_rep_std=None,
                 with_focal_length=False, with_bbox_info=False, with_cam_center=False,
                 scene_feat_dim=512, scene_type="whole_scene", scene_cano=False,
                 weight_loss_v2v=0, weight_loss_keypoints_3d=0, weight_loss_keypoints_3d_full=0, weight_loss_keypoints_2d_full=0,
                 weight_loss_betas=0, weight_loss_body_pose=0, weight_loss_global_orient=0, weight_loss_pose_6d_ortho=0,
                 weight_coap_penetration=0, start_coap_epoch=0, cond_mask_prob=0, only_mask_img_cond=False,
                 diffusion_blk=4, gcn_dropout=0.0, gcn_nonlocal_layer=False, gcn_hid_dim=1024,
                 pelvis_vis_loosen=False, diffuse_fuse=False, smpl_asset=None, smpl_model_path="data/smpl", allow_synthetic_smpl=False):
        super().__init__()
        self.cfg = cfg if cfg is not None else default_cfg()
        self.device = torch.device(device) if device is not None else torch.device("cuda")
        if not with_focal_length:
            # the reference itself cannot be constructed this way: `if self.with_focal_length or self.with_vfov` (egohmr.py:77) reads an
            # attribute that is never set
            raise AttributeError("'EgoHMR' object has no attribute 'with_vfov' (the reference fails the same way for with_focal_length=False, "
                                 "models/egohmr/egohmr.py:77); pass with_focal_length=True")
        self.with_focal_length, self.with_bbox_info, self.with_cam_center = True, bool(with_bbox_info), bool(with_cam_center)
        self.scene_type, self.scene_cano = scene_type, scene_cano
        self.only_mask_img_cond, self.diffuse_fuse = only_mask_img_cond, diffuse_fuse
        # cond_mask_prob only acts under self.training (mask_cond, egohmr.py:159-168): the sampling path is eval-only, so it is kept and ignored
        self.cond_mask_prob = float(cond_mask_prob)
        self.diffuse_feat_dim = 6
        dev = self.device
        self.register_buffer("body_rep_mean_buf", torch.as_tensor(body_rep_mean, dtype=torch.float32).reshape(144).clone(), persistent=False)
        self.register_buffer("body_rep_std_buf", torch.as_tensor(body_rep_std, dtype=torch.float32).reshape(144).clone(), persistent=False)
        self.body_rep_mean, self.body_rep_std = body_rep_mean, body_rep_std

        self.input_process = InputProcess(6, 512)
        self.sequence_pos_encoder = PositionalEncoding(512)
        self.embed_timestep = TimestepEmbedder(512, self.sequence_pos_encoder)
        if self.cfg.MODEL.BACKBONE.NUM_LAYERS != 50:
            raise NotImplementedError("only the ResNet-50 backbone of configs/prohmr.yaml is built")
        self.backbone = ResNet50Features()
        self.scene_enc = ResnetPointnet(out_dim=scene_feat_dim, hidden_dim=256)
        self.transl_enc = TranslEnc(3, 128)
        ctx = self.cfg.MODEL.BACKBONE.OUT_CHANNELS + 1 + (3 if with_bbox_info else 0) + (2 if with_cam_center else 0) + scene_feat_dim + 128   # :74-83
        self.context_feats_dim = ctx
        self.cond_split = (IMG_DIM, ctx, ctx + 512, ctx + 1024)
        self.diffusion_model = ModulatedGCN(adj=smpl_tree_adjacency(), in_dim=ctx + 512 + 512, hid_dim=gcn_hid_dim, out_dim=6,
                                            num_layers=diffusion_blk, nonlocal_layer=gcn_nonlocal_layer)
        self.beta_layer = FCHeadBeta(ctx)
        self.smpl = smpl_mod.create(smpl_model_path, model_type="smpl", gender="neutral", asset=smpl_asset, allow_synthetic=allow_synthetic_smpl)
        self.openpose_to_smpl = OPENPOSE_TO_SMPL_LOOSE if pelvis_vis_loosen else OPENPOSE_TO_SMPL
        self.smpl_to_openpose = [24, 12, 17, 19, 21, 16, 18, 20, 0, 2, 5, 8, 1, 4, 7, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34]
        self.collision_tau = 0.05
        self.guide_reduction = "mean"          # COAP variant: -loss.mean() (egohmr.py:562); 'sum' = VolSMPL variant (egohmr_volsmpl.py:618)
        self.guide_denom_override = None       # sharded / sub-batch runs: the GLOBAL batch size of `-loss.mean()`, else None
        self.guide_all_points = False          # COAP variant: bbox-selected scene points (egohmr.py:550-552); True = all points (egohmr_volsmpl.py:609-612)
        self.lbs_every_step = True             # EgoHMR.forward decodes the body in every step (egohmr.py:276)
        self.prune_passes = True           # exact: items whose 24 joints are all visible skip the image-masked pass (egohmr.py:239-254)
        self.overlap_encoders = True       # ResNet-50 and the scene PointNet on two HIP streams (FusedSampler.prepare)
        self.backbone_matrix_core = True   # ResNet-50 blocks as split-f16 implicit GEMMs (csrc/conv.hip); False = library convs + ehm_bias_act
        # arithmetic of the hidden GCN convs: 'f32' (f32-input MFMA), 'f16x3' (split-f16 MFMA, f32-grade), 'f16' (plain f16, not parity-grade)
        self.gcn_precision = "f16x3"
        # precision schedule: only the LAST k executed steps of a fused sampling loop run in gcn_precision ('f16x3'),
        # the earlier ones on plain f16 operands / f16 activations.  'auto' = max(10, ceil(T / 10)) for T >= 20 (the last tenth of the
        # schedule), ceil(0.4 T) for 10 <= T < 20, off below (errors of earlier steps are contracted away by the posterior mean,
        # measured in tools/precision_schedule.py: final bodies within 7e-6 m of the all-f16x3 run at B=256 for DDPM-100 / DDIM-50 /
        # DDIM-10; the parity bar is 1e-4 m); an int = that k; None = off
        self.f16x3_last_steps = "auto"
        # hipGraph replay of the whole T-step loop (one graph launch instead of ~5 T kernel launches), unguided loops only.  Off by
        # default: measured on MI355X (tools/latency_small.py, profiles/r02_latency_b8_ddim5.json) the loop is GPU-bound even at B = 8
        # (3.74 ms eager vs 3.73 ms replayed for DDIM-5: the eight chained convs of a step are a dependency chain of tile times),
        # so the graph only saves host CPU time.  True = use it; 'auto' = for passes * B <= 64.
        self.use_hip_graph = False
        self.fused_sampler = FusedSampler(self)
        self.to(dev)
        self.eval()

    # ------------------------------------------------------------------ reference API
    def validation_setup(self):
        self.training = False
        self.eval()

    def _std_mean(self):
        return self.body_rep_mean_buf, self.body_rep_std_buf

    def visibility(self, batch):
        vis = batch["orig_keypoints_2d"][:, :, -1] > 0                                 # :186
        vis = vis.clone()
        vis[:, 8] = True                                                               # :187
        return vis[:, self.openpose_to_smpl]                                           # :188

    def forward(self, batch, timesteps, eval_with_uncond=True):
        """One denoising evaluation (egohmr.py:173-303).  Conditioning is cached per batch object."""
        fs = self.fused_sampler
        st = fs.prepare(batch)
        x_t = _lib.f32(batch["x_t"], self.device).reshape(-1, 144)
        B = x_t.shape[0]
        passes = 2 if (self.diffuse_fuse and eval_with_uncond) else 1
        tvec = fs.timestep_vectors(timesteps[:1].to(self.device))[0]
        x0 = fs.denoise_once(st, x_t, tvec, passes)
        mean, std = self._std_mean()
        verts = torch.empty(B, self.smpl.num_verts, 3, device=self.device)
        joints = torch.empty(B, self.smpl.num_joints_out, 3, device=self.device)
        R = torch.empty(B, 24, 3, 3, device=self.device)
        pose6d = torch.empty(B, 144, device=self.device)
        _lib.check(_lib.lib().ehm_smpl_forward_rot6d(self.smpl.handle(), _lib.ptr(st.betas), _lib.ptr(x0), _lib.ptr(mean), _lib.ptr(std),
                                                     _lib.ptr(verts), _lib.ptr(joints), _lib.ptr(R), _lib.ptr(pose6d), None, B,
                                                     _lib.stream_ptr()), "ehm_smpl_forward_rot6d")
        batch["vis_mask_smpl"] = st.vis_bool
        return self._pack_output(batch, st, x0, pose6d, R, verts, joints)

    def _pack_output(self, batch, st, x0, pose6d, R, verts, joints):
        self.scene_pcd_verts = st.scene
        self.input_transl = st.transl
        self.smpl_output = smpl_mod.SMPLOutput(vertices=verts, joints=joints, full_pose=R)
        focal = st.fx.unsqueeze(-1).repeat(1, 2) * self.cfg.CAM.FX_NORM_COEFF          # :283-285
        center = torch.stack([st.cam_cx, st.cam_cy], dim=-1)
        self.focal_length, self.camera_center_full = focal, center
        kp2d = geometry.perspective_projection(joints, st.transl, focal, center)        # :295-298
        kp2d = torch.stack([kp2d[..., 0] / 1920 - 0.5, kp2d[..., 1] / 1080 - 0.5], dim=-1)
        return {
            "pred_x_start": x0,
            "pred_smpl_params": {"global_orient": R[:, [0]].clone(), "body_pose": R[:, 1:].clone(), "betas": st.betas.clone()},
            "pred_pose_6d": pose6d,
            "pred_keypoints_3d": joints,
            "pred_vertices": verts,
            "pred_keypoints_3d_full": joints + st.transl.unsqueeze(1),
            "pred_keypoints_2d_full": kp2d,
        }

    def guide_coll(self, batch, output, t, compute_grad="x_t"):
        """egohmr.py:517-570 with the build's collision proxy in place of COAP; returns [B,144]."""
        fs = self.fused_sampler
        st = fs.prepare(batch)
        x = _lib.f32(batch["x_t"] if compute_grad == "x_t" else output["pred_x_start"], self.device).reshape(-1, 144)
        return fs.guidance_gradient(st, x, _lib.f32(output["pred_smpl_params"]["betas"], self.device))

    def eval_coll(self, output):
        """egohmr.py:487-514 with the build's proxy in place of `coap.query(...) > 0.5`: per item, the share of the scene points that
        lie inside the body's bounding box AND closer than tau to its surface.  One kernel sequence for the batch, one host sync
        (`.tolist()`, the reference syncs per item); returns the reference's python list [B] of floats.  NOT a COAP number."""
        p = output["pred_smpl_params"]
        so = self.smpl(betas=p["betas"], body_pose=p["body_pose"], global_orient=p["global_orient"], pose2rot=False)
        _, _, hits = self.fused_sampler.collision(so.vertices, self.scene_pcd_verts, want_grad=False, want_hits=True)
        return (hits.float() / self.scene_pcd_verts.shape[1]).tolist()

    def compute_loss(self, batch, output, cur_epoch=0):
        """Evaluation losses need ground-truth annotations (egohmr.py:307-449); the sampling path has none."""
        output["losses"] = {}
        return torch.zeros((), device=self.device)

    def training_step(self, *a, **k):
        raise NotImplementedError("training is outside the sampling hot path this package implements")


class EgoHMRVolsmpl(EgoHMR):
    """models/egohmr/egohmr_volsmpl.py: the same network with VolumetricSMPL instead of COAP behind the collision guidance.
    What differs on the sampling path (everything else in that file is a re-formatted copy of egohmr.py):
      guide_coll          :582-629  one BATCHED `volume.collision_loss(scene, smpl_output)` over ALL scene points (no per-item bounding-box
                                    selection), gradient of `-loss.sum()` (no 1/B factor); default guidance weight 30 (test_egohmr_volsmpl.py:62)
      eval_coll_volsmpl   :548-579  per item, bbox-selected points with `volume.query_fast(...) < 0` over N
      eval_coll           :519-546  the COAP metric, kept (the reference keeps COAP attached for training)
    VolumetricSMPL is a learned SDF that cannot be obtained offline: the collision term is the build's proxy,
    with `sdf < 0` read as `distance to the surface < tau`.  Numbers from it are NOT VolumetricSMPL numbers."""

    DEFAULT_COND_GRAD_WEIGHT = 30.0            # test_egohmr_volsmpl.py:62

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.guide_reduction = "sum"
        self.guide_all_points = True

    def eval_coll_volsmpl(self, output):
        p = output["pred_smpl_params"]
        so = self.smpl(betas=p["betas"], body_pose=p["body_pose"], global_orient=p["global_orient"], pose2rot=False)
        _, _, hits = self.fused_sampler.collision(so.vertices, self.scene_pcd_verts, want_grad=False, want_hits=True, all_points=False)
        return (hits.float() / self.scene_pcd_verts.shape[1]).tolist()


# ---------------------------------------------------------------------------------------------- native engine
class _Prepared(SimpleNamespace):
    pass


class FusedSampler:
    """Owns the native denoiser handle and runs sampling loops through ehm_sample_loop."""

    def __init__(self, model: EgoHMR):
        self._model_ref = [model]
        self._gcn = None
        self._gcn_key = None
        self._folded = None
        self._prep_key = None
        self._prep = None
        self._ws = None
        self._graphs = {}
        self.last_trace = None

    @property
    def model(self) -> EgoHMR:
        return self._model_ref[0]

    # ------------------------------------------------------------------ weights -> native handle
    def _param_key(self):
        if getattr(self, "_pk", None) is None:
            self._pk = _lib.TensorKey(self.model.diffusion_model, self.model.input_process)
        return self._pk()

    def gcn(self):
        key = self._param_key()
        if self._gcn is None or key != self._gcn_key:
            self._free()
            m = self.model
            dm = m.diffusion_model
            keep = []

            def params(gc, bn):
                def t(x):
                    x = _lib.f32(x, m.device)
                    keep.append(x)
                    return x.data_ptr()
                p = _lib.GConvParams()
                p.W, p.M, p.adj2, p.bias = t(gc.W), t(gc.M), t(gc.adj2), t(gc.bias)
                if bn is not None:
                    p.bn_weight, p.bn_bias, p.bn_mean, p.bn_var = t(bn.weight), t(bn.bias), t(bn.running_mean), t(bn.running_var)
                p.in_dim, p.out_dim = gc.in_features, gc.out_features
                return p

            gi = dm.gconv_input[0]
            inp = params(gi.gconv, gi.bn)
            hidden = []
            for blk in dm.gconv_layers:
                hidden += [params(blk.gconv1.gconv, blk.gconv1.bn), params(blk.gconv2.gconv, blk.gconv2.bn)]
            outp = params(dm.gconv_output, None)
            arr = (_lib.GConvParams * len(hidden))(*hidden)
            adj = _lib.f32(dm.adj, m.device)
            h = C.c_void_p()
            with torch.cuda.device(m.device):
                _lib.check(_lib.lib().ehm_gcn_create(C.byref(h), _lib.ptr(adj), C.byref(inp), arr, len(hidden), C.byref(outp), dm.hid_dim,
                                                     _lib.stream_ptr()), "ehm_gcn_create")
            self._gcn, self._gcn_key = h, key
            # fold InputProcess (Linear 6->512) into the x_t slice of the input conv: x @ (Wp^T W_k[2694:3206]) + bp W_k[...]
            W = gi.gconv.W.detach().double()                                            # [2, 3718, hid]
            Wp, bp = m.input_process.poseEmbedding.weight.detach().double(), m.input_process.poseEmbedding.bias.detach().double()
            a, b, c, d = m.cond_split
            Wx = torch.einsum("ec,kef->kcf", Wp, W[:, b:c, :])                          # [2,6,hid]
            bx = torch.einsum("e,kef->kf", bp, W[:, b:c, :])                            # [2,hid]
            # the image / scene+translation+camera slices as ONE [K, 2*hid] matrix each (both branches side by side, K padded to 32 with
            # zero rows): operands of ehm_skinny_gemm_f32 in prepare()
            hid = dm.hid_dim
            Wd = gi.gconv.W.detach().float()
            k_oth = (b - a + 31) // 32 * 32
            W_img_cat = Wd[:, :a, :].permute(1, 0, 2).reshape(a, 2 * hid).contiguous()
            W_oth_cat = torch.zeros(k_oth, 2 * hid, device=m.device)
            W_oth_cat[:b - a] = Wd[:, a:b, :].permute(1, 0, 2).reshape(b - a, 2 * hid)
            self._folded = SimpleNamespace(Wx=Wx.float().contiguous(), bx=bx, W_img=gi.gconv.W.detach()[:, :a, :],
                                           W_oth=gi.gconv.W.detach()[:, a:b, :], W_t=W[:, c:d, :], W_img_cat=W_img_cat, W_oth_cat=W_oth_cat,
                                           k_oth=k_oth)
        _lib.check(_lib.lib().ehm_gcn_set_uncond_mode(self._gcn, 0 if self.model.only_mask_img_cond else 1), "ehm_gcn_set_uncond_mode")
        mode = PRECISIONS[self.model.gcn_precision]
        if _lib.lib().ehm_gcn_get_precision(self._gcn) != mode:
            _lib.check(_lib.lib().ehm_gcn_set_precision(self._gcn, mode), "ehm_gcn_set_precision")
        return self._gcn

    def _free(self):
        if self._gcn is not None:
            try:
                _lib.lib().ehm_gcn_destroy(self._gcn)
            except Exception:
                pass
            self._gcn = None

    def __del__(self):
        self._free()

    def _backbone_fn(self):
        """ResNet-50 with BatchNorm folded into the convolutions, rebuilt when the backbone weights change."""
        bb = self.model.backbone
        if getattr(self, "_bbk", None) is None or self._bbk.modules[0] is not bb:
            self._bbk = _lib.TensorKey(bb)
        key = self._bbk()
        if getattr(self, "_bb_key", None) != key:
            self._bb_fn, self._bb_key = bb.folded(channels_last=False, matrix_core=self.model.backbone_matrix_core), key
        return self._bb_fn

    # ------------------------------------------------------------------ step-invariant conditioning
    @torch.no_grad()
    def prepare(self, batch) -> _Prepared:
        """Everything in EgoHMR.forward that does not depend on x_t / t (egohmr.py:182-223, :263-265)."""
        m = self.model
        # Cache key = identity AND version of every tensor the conditioning is computed from, and of every weight it passes
        # through.  The cached entry keeps strong references to those input tensors, so neither their id() nor their storage can be
        # recycled for a different batch while the entry is alive; in-place edits bump _version.
        ins = [batch["img"], batch["scene_pcd_verts_full"], batch["orig_keypoints_2d"], batch["fx"], batch["cam_cx"], batch["cam_cy"],
               batch["box_center"], batch["box_size"], batch["smpl_params"]["transl"]]
        key = tuple((id(t), t._version, t.data_ptr()) for t in ins) + self._param_key() + self._cond_param_key()
        if self._prep is not None and self._prep_key == key:
            return self._prep
        self.gcn()
        dev = m.device
        g = lambda k: _lib.f32(batch[k], dev)
        transl = _lib.f32(batch["smpl_params"]["transl"], dev)
        scene = g("scene_pcd_verts_full")
        if m.scene_cano:
            scene = scene - transl.unsqueeze(1)                                        # :211
        scene = scene.contiguous()
        img = g("img")
        # pass pruning map (ehm_gcn_set_pass_map): items with an invisible joint need the second pass.  Its count is the ONE host
        # read-back of a batch.  It is REQUESTED first (fixed-size device ops + an asynchronous copy into pinned memory + an event) and
        # LOOKED AT after the encoders have been enqueued: in a pipeline of batches the copy sits behind the previous batch's sampling
        # loop, the host spends that time enqueuing this batch's encoders, and when the event fires the GPU walks straight into them
        # while the host enqueues the loop - the GPU never waits for Python.  (Read back at the end of prepare() with a blocking
        # nonzero(), the GPU idled while Python built the step table: -6 %; read back with a blocking nonzero() at the top, it idled
        # ~0.5 ms per batch until the first encoder kernels arrived: same-box A/B 3493 / 3505 -> 3504 / 3506 bodies/s, DDIM-10 +1-3 %.)
        vis = m.visibility({"orig_keypoints_2d": batch["orig_keypoints_2d"].to(dev)})
        need = ~vis.all(dim=1)
        order = torch.argsort((~need).to(torch.uint8), stable=True).to(torch.int32)        # items that need the second pass first, ascending
        mask_slot = torch.where(need, torch.cumsum(need.to(torch.int32), 0) - 1, torch.full_like(need, -1, dtype=torch.int32)).to(torch.int32).contiguous()
        if getattr(self, "_count_host", None) is None:
            self._count_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._count_host.copy_(need.sum(dtype=torch.int32).reshape(1), non_blocking=True)
        count_ready = torch.cuda.Event()
        count_ready.record(torch.cuda.current_stream(dev))
        # The two encoders are independent, and complementary on the chip: ResNet-50's early layers stream 0.8 GB float32 activations
        # per conv (HBM-bound, matrix cores idle), the PointNet's GEMMs are matrix-core bound.  Run them on two HIP streams.
        if m.overlap_encoders:
            cur = torch.cuda.current_stream(dev)
            if getattr(self, "_side_stream", None) is None or self._side_stream.device != dev:
                self._side_stream = torch.cuda.Stream(device=dev)
            side = self._side_stream
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                scene_feats = m.scene_enc(scene)                                       # :214
            img_feats = self._backbone_fn()(img)                                       # :183 (BatchNorm folded into the convs)
            cur.wait_stream(side)
            scene_feats.record_stream(cur)
            scene.record_stream(side)
        else:
            img_feats = self._backbone_fn()(img)
            scene_feats = m.scene_enc(scene)
        transl_feat = m.transl_enc(transl)                                             # :217
        fx, cx, cy = g("fx"), g("cam_cx"), g("cam_cy")
        ofx = fx * m.cfg.CAM.FX_NORM_COEFF
        bc, bs = g("box_center"), g("box_size")
        cam = [fx.unsqueeze(1)]                                                        # :195-205 (prepended in this order)
        if m.with_bbox_info:
            cam = [torch.stack([bc[:, 0] / ofx, bc[:, 1] / ofx, bs / ofx], -1)] + cam
        if m.with_cam_center:
            cam = [torch.stack([cx / ofx, cy / ofx], -1)] + cam
        cam = torch.cat(cam, dim=1)
        other = torch.cat([scene_feats, transl_feat, cam], dim=1)                      # :220-221
        f = self._folded
        h_img, h_oth, betas = self._project(img_feats.contiguous(), other)
        count_ready.synchronize()
        num_masked = int(self._count_host[0])
        mask_items = order[:num_masked].contiguous()
        self._prep = _Prepared(B=img_feats.shape[0], h_img=h_img, h_oth=h_oth, vis=vis.to(torch.uint8).contiguous(), vis_bool=vis,
                               betas=betas, scene=scene, transl=transl, fx=fx, cam_cx=cx, cam_cy=cy, img_feats=img_feats,
                               scene_feats=scene_feats)
        self._prep.inputs = ins                  # strong references (see the key above)
        self._prep.mask_items, self._prep.mask_slot, self._prep.num_masked = mask_items, mask_slot, num_masked
        self._prep_key = key
        return self._prep

    def _project(self, img_feats, other):
        """The step-invariant slices of the input graph conv ([B,2,hid] each: image features, scene + translation + camera features)
        and the beta head (egohmr.py:263-265) as exact-float32 matrix-core GEMMs built for M = B rows (ehm_skinny_gemm_f32)."""
        m, f, L = self.model, self._folded, _lib.lib()
        B, dev, hid = img_feats.shape[0], img_feats.device, m.diffusion_model.hid_dim
        st = _lib.stream_ptr()
        if img_feats.shape[1] % 32:
            raise _lib.EgoHMRHipError(f"image feature width {img_feats.shape[1]} is not a multiple of 32")
        oth = torch.zeros(B, f.k_oth, device=dev)
        oth[:, :other.shape[1]] = other
        h_img = torch.empty(B, 2, hid, device=dev)
        h_oth = torch.empty(B, 2, hid, device=dev)
        _lib.check(L.ehm_skinny_gemm_f32(_lib.ptr(img_feats), _lib.ptr(f.W_img_cat), None, _lib.ptr(h_img), B, img_feats.shape[1], 2 * hid, 0, st), "ehm_skinny_gemm_f32")
        _lib.check(L.ehm_skinny_gemm_f32(_lib.ptr(oth), _lib.ptr(f.W_oth_cat), None, _lib.ptr(h_oth), B, f.k_oth, 2 * hid, 0, st), "ehm_skinny_gemm_f32")
        # beta head: Linear(2048 + 646 -> 1024) + ReLU on the same kernel (weights transposed and padded once per weight version), the
        # 1024 -> 10 layer and init_betas in torch
        l1, l2 = m.beta_layer.layers[0], m.beta_layer.layers[2]
        key = (l1.weight.data_ptr(), l1.weight._version, l1.bias.data_ptr(), l1.bias._version, str(dev))
        if getattr(self, "_beta_key", None) != key:
            a = img_feats.shape[1]
            Wt = torch.zeros(a + f.k_oth, l1.out_features, device=dev)
            w = l1.weight.detach().float().to(dev)
            Wt[:a] = w[:, :a].t()
            Wt[a:a + other.shape[1]] = w[:, a:].t()
            self._beta_w1, self._beta_b1, self._beta_key = Wt.contiguous(), l1.bias.detach().float().to(dev).contiguous(), key
        if l1.out_features % 32 == 0 and l1.in_features == img_feats.shape[1] + other.shape[1]:
            xb = torch.cat([img_feats, oth], dim=1)
            hb = torch.empty(B, l1.out_features, device=dev)
            _lib.check(L.ehm_skinny_gemm_f32(_lib.ptr(xb), _lib.ptr(self._beta_w1), _lib.ptr(self._beta_b1), _lib.ptr(hb), B, xb.shape[1], l1.out_features, 1, st),
                       "ehm_skinny_gemm_f32")
            betas = (l2(hb) + m.beta_layer.init_betas).contiguous()
        else:
            betas = m.beta_layer(torch.cat([img_feats, other], dim=1)).contiguous()
        return h_img, h_oth, betas

    def _cond_param_key(self):
        m = self.model
        if getattr(self, "_ck", None) is None:
            self._ck = _lib.TensorKey(m.backbone, m.scene_enc, m.transl_enc, m.beta_layer, m.embed_timestep)
        return self._ck()

    def _apply_pass_map(self, st, passes):
        """(virtual bodies, num_masked for the descriptor) after telling the handle which items still need the second pass."""
        m, L = self.model, _lib.lib()
        h = self.gcn()
        if passes == 2 and m.prune_passes:
            _lib.check(L.ehm_gcn_set_pass_map(h, _lib.ptr(st.mask_items) if st.num_masked else None, _lib.ptr(st.mask_slot), st.num_masked), "ehm_gcn_set_pass_map")
            return st.B + st.num_masked, st.num_masked
        _lib.check(L.ehm_gcn_set_pass_map(h, None, None, -1), "ehm_gcn_set_pass_map")
        return passes * st.B, -1

    def invalidate(self, structure: bool = False):
        """Drop the cached conditioning (bench.py: the encoders are part of every timed call).  structure=True also re-collects the
        parameter slots behind the weight-version keys (needed only after sub-modules or parameters were ADDED to the model)."""
        self._prep, self._prep_key = None, None
        if structure:
            self._pk = self._ck = self._bbk = None

    @torch.no_grad()
    def timestep_vectors(self, t_orig: torch.Tensor) -> torch.Tensor:
        """[n] original timesteps -> [n,2,hid]: TimestepEmbedder (egohmr.py:642-643) pushed through the timestep
        slice of the input conv, plus the folded InputProcess bias."""
        m = self.model
        self.gcn()
        temb = m.embed_timestep.time_embed(m.sequence_pos_encoder.pe[t_orig][:, 0])    # [n,512]
        tv = torch.einsum("ne,kef->nkf", temb.double(), self._folded.W_t) + self._folded.bx[None]
        return tv.float().contiguous()

    # ------------------------------------------------------------------ granular denoiser (EgoHMR.forward)
    def _workspace(self, nbytes, device):
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != device:
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        return self._ws

    @torch.no_grad()
    def denoise_once(self, st, x_t, tvec, passes):
        m, L = self.model, _lib.lib()
        hid, B = m.diffusion_model.hid_dim, st.B
        tile = L.ehm_gcn_row_tile()
        rows = self._apply_pass_map(st, passes)[0] * 24
        rows_pad = (rows + tile - 1) // tile * tile
        X = [torch.zeros(rows_pad, hid, device=m.device) for _ in range(3)]
        s = _lib.stream_ptr()
        h = self.gcn()
        _lib.check(L.ehm_gcn_input_layer(h, _lib.ptr(st.h_img), _lib.ptr(st.h_oth), _lib.ptr(st.vis), _lib.ptr(x_t), _lib.ptr(self._folded.Wx),
                                         _lib.ptr(tvec), _lib.ptr(X[0]), B, passes, s), "ehm_gcn_input_layer")
        bufs = (C.c_void_p * 3)(*[x.data_ptr() for x in X])
        res = C.c_int(0)
        _lib.check(L.ehm_gcn_hidden_stack(h, bufs, rows_pad, C.byref(res), s), "ehm_gcn_hidden_stack")
        cur = res.value
        feat = X[cur]
        if m.diffusion_model.nonlocal_layer:
            if m.gcn_precision == "f16":
                raise _lib.EgoHMRHipError("the optional non-local GCN block runs on float32 features; use gcn_precision 'f16x3' or 'f32' with it")
            feat = self._non_local(feat, rows, rows_pad)
        x0 = torch.empty(B, 144, device=m.device)
        _lib.check(L.ehm_gcn_output_layer(h, _lib.ptr(feat), _lib.ptr(st.vis), _lib.ptr(x0), B, passes, s), "ehm_gcn_output_layer")
        self.last_hidden = feat[:rows]
        return x0

    @torch.no_grad()
    def _non_local(self, X, rows, rows_pad):
        """NONLocalBlock2D on the joint axis (modulated_gcn.py:104-110): [theta|phi|g] as ONE 1x1-conv GEMM and W + BatchNorm(eval,
        folded) + residual as another, both on ehm_conv_nhwc_split (rows = N, H = W = 1); the 24 x 24 softmax attention per body
        in ehm_nonlocal_attention."""
        import math
        m, L = self.model, _lib.lib()
        nl = m.diffusion_model.non_local
        hid, ci = m.diffusion_model.hid_dim, nl.inter_channels
        key = tuple((p.data_ptr(), p._version) for p in list(nl.parameters()) + list(nl.buffers()))
        if getattr(self, "_nl_key", None) != key:
            def pack(w2, bias):                                   # [Co, K] float32 -> X2 split weights for the conv kernel
                Co, K = w2.shape
                Co_pad = (Co + 127) // 128 * 128
                wp = torch.zeros(Co_pad, K, device=m.device)
                wp[:Co] = w2
                amax = float(wp.abs().max())
                scale = 2.0 ** math.floor(math.log2(2048.0 / amax)) if amax > 0 else 1.0
                buf = torch.empty(Co_pad, K, device=m.device)
                _lib.check(L.ehm_split_pack(wp.data_ptr(), buf.data_ptr(), Co_pad, K, K, scale, _lib.stream_ptr()), "ehm_split_pack")
                return buf, scale, bias.float().contiguous()
            wqkv = torch.cat([nl.theta.weight, nl.phi.weight, nl.g.weight], 0).flatten(1).float()
            bqkv = torch.cat([nl.theta.bias, nl.phi.bias, nl.g.bias], 0)
            bn = nl.W[1]
            sc = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
            ww = (nl.W[0].weight.flatten(1).double() * sc[:, None]).float()
            bw = ((nl.W[0].bias.double() - bn.running_mean.double()) * sc + bn.bias.double()).float()
            self._nl_packed, self._nl_key = (pack(wqkv, bqkv), pack(ww, bw)), key
        (wq, sq, bq), (wo, so, bo) = self._nl_packed
        s = _lib.stream_ptr()
        qkv = torch.empty(rows, 3 * ci, device=m.device)
        d = _lib.ConvDesc(X.data_ptr(), wq.data_ptr(), bq.data_ptr(), None, qkv.data_ptr(), rows, 1, 1, hid, 3 * ci, 1, 1, 1, 0, 0, sq)
        _lib.check(L.ehm_conv_nhwc_split(C.byref(d), s), "ehm_conv_nhwc_split")
        y = torch.empty(rows, ci, device=m.device)
        _lib.check(L.ehm_nonlocal_attention(qkv.data_ptr(), y.data_ptr(), rows // 24, ci, s), "ehm_nonlocal_attention")
        Z = torch.zeros(rows_pad, hid, device=m.device)
        d = _lib.ConvDesc(y.data_ptr(), wo.data_ptr(), bo.data_ptr(), X.data_ptr(), Z.data_ptr(), rows, 1, 1, ci, hid, 1, 1, 1, 0, 0, so)
        _lib.check(L.ehm_conv_nhwc_split(C.byref(d), s), "ehm_conv_nhwc_split")
        return Z

    # ------------------------------------------------------------------ guidance pieces
    @torch.no_grad()
    def collision(self, verts, scene, want_grad=True, want_hits=False, all_points=None):
        """The collision proxy for a batch of bodies: (loss [B], d loss / d verts [B,V,3] or None, hits [B] int32 or None)."""
        m, L = self.model, _lib.lib()
        verts, scene = _lib.f32(verts, m.device), _lib.f32(scene, m.device)
        B, V, N = verts.shape[0], verts.shape[1], scene.shape[1]
        loss = torch.empty(B, device=m.device)
        gverts = torch.empty_like(verts) if want_grad else None
        hits = torch.empty(B, device=m.device, dtype=torch.int32) if want_hits else None
        allp = m.guide_all_points if all_points is None else all_points
        with torch.cuda.device(m.device):
            _lib.check(L.ehm_collision_query(_lib.ptr(verts), _lib.ptr(scene), _lib.ptr(loss), _lib.ptr(gverts), _lib.ptr(hits), B, V, N,
                                             m.collision_tau, int(bool(allp)), _lib.stream_ptr()), "ehm_collision_query")
        return loss, gverts, hits

    @torch.no_grad()
    def guidance_gradient(self, st, x, betas):
        m, L = self.model, _lib.lib()
        B = x.shape[0]
        mean, std = m._std_mean()
        verts = torch.empty(B, m.smpl.num_verts, 3, device=m.device)
        joints = torch.empty(B, m.smpl.num_joints_out, 3, device=m.device)
        s = _lib.stream_ptr()
        _lib.check(L.ehm_smpl_forward_rot6d(m.smpl.handle(), _lib.ptr(betas), _lib.ptr(x), _lib.ptr(mean), _lib.ptr(std), _lib.ptr(verts),
                                            _lib.ptr(joints), None, None, None, B, s), "ehm_smpl_forward_rot6d")
        loss, gverts, _ = self.collision(verts, st.scene)
        gpose = torch.empty(B, 144, device=m.device)
        _lib.check(L.ehm_smpl_backward_rot6d(m.smpl.handle(), _lib.ptr(betas), _lib.ptr(x), _lib.ptr(mean), _lib.ptr(std), _lib.ptr(gverts),
                                             _lib.ptr(gpose), B, s), "ehm_smpl_backward_rot6d")
        grad = torch.empty(B, 144, device=m.device)
        denom = self.guide_denom(B)
        _lib.check(L.ehm_guidance_grad_finish(_lib.ptr(gpose), _lib.ptr(loss), _lib.ptr(grad), B, denom, s), "ehm_guidance_grad_finish")
        return grad

    def guide_denom(self, B: int) -> float:
        """Denominator of the guidance gradient: B for `-loss.mean()` (egohmr.py:562), 1 for `-loss.sum()` (egohmr_volsmpl.py:618)."""
        m = self.model
        if m.guide_reduction != "mean":
            return 1.0
        return float(m.guide_denom_override) if m.guide_denom_override else float(B)

    def lowprec_steps(self, T: int, guided=False, ddim: bool = True) -> int:
        """How many LEADING steps of a T-step fused loop run on plain f16 operands (EgoHMR.f16x3_last_steps).  `guided` = number of
        collision-guided steps at the END of the loop (True = unknown).  The guidance feeds nearest-vertex switches back with gain, so
        the f16 steps must end well before the first guided one: the 20 steps ahead of it also run in f16x3.  Measured
        (tools/precision_schedule.py --guided, profiles/r02_precision_schedule_ddpm100_guided_b128.jsonl): DDPM-100 at B=128 stays
        within 1.3e-5 m of the all-f16x3 run for every k >= 12 (1e-6 for two seeds of three; the third has one body at a
        nearest-vertex switch and shows the same 1.2e-5 at k = 60), and x_t inside the guided steps within 1.4e-4 at k = 30; on the
        DDPM-50 guided golden the guided-step trace moves by 3.5e-3 at k = 20 and by 2.2e-4 at k = 30 (final vertices 6e-6 / 2e-6)."""
        k = self.model.f16x3_last_steps
        if k is None or self.model.gcn_precision != "f16x3":
            return 0
        if k == "auto":
            if guided is True:
                return 0
            n_guided = int(guided)
            if T < 10:
                return 0                         # (not measured below ten steps)
            if T < 20:
                k = -(-2 * T // 5)               # short loops: the last 40 % (DDIM-10: k = 4 -> <= 5.1e-6 m, k = 5 -> <= 4.6e-6, k = 3 -> <= 7.4e-6; 3 seeds x 2 respacings)
            elif ddim:
                k = max(10, -(-T // 10))         # DDIM-50: k = 10 -> 6.5e-6 m
            else:
                k = max(8, -(-2 * T // 25))      # ancestral sampling contracts harder: DDPM-100 k = 8 -> <= 3.9e-6 m over 4 seeds (k = 5: 7.4e-6)
            if n_guided:
                k = max(k, n_guided + 20)
        return max(0, T - int(k))

    # ------------------------------------------------------------------ S samples of one batch in ONE loop
    @torch.no_grad()
    def run_samples(self, diffusion, batch, noise_stacks, ddim=False, guided=False, cond_grad_weight=1.0, defer_status=False):
        """The reference draws S samples per item with S sequential sampling loops over the same batch (test_egohmr.py:251-266).  The
        samples are independent given the conditioning, so this runs them as ONE loop over S*B bodies (sample-major: body s*B + b) with
        the conditioning replicated by index - the same arithmetic per body (the guidance denominator stays B), S times fewer launches
        and full-size conv tiles for small B.  noise_stacks: S tensors [T+1,B,144].  Returns a list of S result dicts like run()."""
        S = len(noise_stacks)
        st = self.prepare(batch)
        if S == 1:
            return [self.run(diffusion, batch, noise_stacks[0], ddim=ddim, guided=guided, cond_grad_weight=cond_grad_weight, prepared=st,
                             defer_status=defer_status)]
        B = st.B
        rep = lambda t: t.repeat(S, *([1] * (t.dim() - 1))).contiguous()
        fields = {k: (rep(v) if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == B and k not in ("mask_items", "mask_slot") else v)
                  for k, v in vars(st).items()}
        r = _Prepared(**fields)
        r.B = S * B
        off = torch.arange(S, device=st.mask_slot.device, dtype=torch.int32)
        nm = max(st.num_masked, 0)
        r.mask_items = (st.mask_items.view(1, -1) + off.view(-1, 1) * B).reshape(-1).contiguous()
        r.mask_slot = torch.where(st.mask_slot.view(1, -1) >= 0, st.mask_slot.view(1, -1) + off.view(-1, 1) * nm,
                                  torch.full((1, 1), -1, device=off.device, dtype=torch.int32)).reshape(-1).to(torch.int32).contiguous()
        r.num_masked = S * st.num_masked if st.num_masked >= 0 else st.num_masked
        r.inputs = st.inputs
        T = diffusion.num_timesteps
        noise = torch.cat([_lib.f32(n, self.model.device)[: T + 1] for n in noise_stacks], dim=1)
        res = self.run(diffusion, dict(batch), noise, ddim=ddim, guided=guided, cond_grad_weight=cond_grad_weight, prepared=r, denom_items=B,
                       defer_status=defer_status)

        def split(x):
            if torch.is_tensor(x):
                return list(x.reshape(S, B, *x.shape[1:]).unbind(0)) if x.dim() >= 1 and x.shape[0] == S * B else [x] * S
            if isinstance(x, dict):
                parts = {k: split(v) for k, v in x.items()}
                return [{k: parts[k][i] for k in x} for i in range(S)]
            return [x] * S
        outs = split(res)
        # leave the model's per-call attributes as S sequential calls would: un-replicated inputs, the last sample's bodies
        m = self.model
        m.scene_pcd_verts, m.input_transl = st.scene, st.transl
        m.focal_length, m.camera_center_full = m.focal_length[:B], m.camera_center_full[:B]
        last = outs[-1]["other_outputs"]
        m.smpl_output = smpl_mod.SMPLOutput(vertices=last["pred_vertices"], joints=last["pred_keypoints_3d"],
                                            full_pose=torch.cat([last["pred_smpl_params"]["global_orient"], last["pred_smpl_params"]["body_pose"]], dim=1))
        return outs

    # ------------------------------------------------------------------ whole loop
    @torch.no_grad()
    def run(self, diffusion, batch, noise_stack, ddim=False, guided=False, cond_grad_weight=1.0, trace=False, prepared=None, denom_items=None,
            defer_status=False):
        """p_sample_loop / ddim_sample_loop (gaussian_diffusion.py:391-508 / :618-718) in one native call.
        Returns the reference's dict(sample, pred_xstart, other_outputs)."""
        m, L = self.model, _lib.lib()
        if m.diffusion_model.nonlocal_layer:
            raise _lib.EgoHMRHipError("the one-call sampling loop does not carry the optional non-local GCN block; "
                                      "use GaussianDiffusion.p_sample_loop / ddim_sample_loop (they take the step-wise route for such a model)")
        ev = getattr(self, "_status_event", None)
        if ev is not None and ev.query():                 # a deferred status word of an earlier call has arrived: look at it now
            self.check_status()
        st = prepared if prepared is not None else self.prepare(batch)
        B, T, hid, V = st.B, diffusion.num_timesteps, m.diffusion_model.hid_dim, m.smpl.num_verts
        noise = _lib.f32(noise_stack, m.device)
        assert noise.shape[0] >= T + 1 and noise.shape[1] == B and noise.shape[2] == 144, noise.shape
        steps = (_lib.StepCoefs * T)(*[diffusion.step_coefs(i, ddim, 0.0, cond_grad_weight, guided) for i in range(T - 1, -1, -1)])
        any_guided = any(s.grad_scale != 0.0 for s in steps)
        first_guided = next((i for i, s in enumerate(steps) if s.grad_scale != 0.0), T)
        n_guided = T - first_guided                       # (the reference guides a contiguous tail: t < 10, gaussian_diffusion.py:378-385)
        tmap = torch.tensor([diffusion.timestep_map[i] for i in range(T - 1, -1, -1)], device=m.device, dtype=torch.long)
        tvecs = self.timestep_vectors(tmap)                                            # [T,2,hid]
        passes = 2 if m.diffuse_fuse else 1
        _, num_masked = self._apply_pass_map(st, passes)
        desc = _lib.SampleDesc(B=B, passes=passes, num_steps=T, ddim=int(ddim),
                               lbs_every_step=int(m.lbs_every_step), num_scene_points=st.scene.shape[1] if any_guided else 0,
                               guide_denom=self.guide_denom(denom_items or B), tau=m.collision_tau, num_masked=num_masked,
                               guide_all_points=int(bool(m.guide_all_points)), lowprec_steps=self.lowprec_steps(T, n_guided, ddim))
        nbytes = L.ehm_sample_workspace_bytes(C.byref(desc), hid, V)
        if nbytes < 0:
            raise _lib.EgoHMRHipError(f"ehm_sample_workspace_bytes rejected the descriptor (rc={nbytes})")
        dev = m.device
        mean, std = m._std_mean()
        gcn, smpl_h = self.gcn(), m.smpl.handle()

        def launch(bufs, ws, tr):
            _lib.check(L.ehm_sample_loop(gcn, smpl_h, C.byref(desc), steps, _lib.ptr(bufs.h_img), _lib.ptr(bufs.h_oth), _lib.ptr(bufs.vis),
                                         _lib.ptr(self._folded.Wx), _lib.ptr(bufs.tvecs), _lib.ptr(bufs.noise),
                                         _lib.ptr(bufs.scene) if any_guided else None, _lib.ptr(bufs.betas), _lib.ptr(mean), _lib.ptr(std),
                                         _lib.ptr(bufs.x_final), _lib.ptr(bufs.x0), _lib.ptr(bufs.verts), _lib.ptr(bufs.joints), _lib.ptr(bufs.R),
                                         _lib.ptr(bufs.pose6d), _lib.ptr(tr), _lib.ptr(ws), nbytes, _lib.stream_ptr()), "ehm_sample_loop")

        def out_bufs():
            return dict(x_final=torch.empty(B, 144, device=dev), x0=torch.empty(B, 144, device=dev), verts=torch.empty(B, V, 3, device=dev),
                        joints=torch.empty(B, m.smpl.num_joints_out, 3, device=dev), R=torch.empty(B, 24, 3, 3, device=dev),
                        pose6d=torch.empty(B, 144, device=dev))

        ins = dict(h_img=st.h_img, h_oth=st.h_oth, vis=st.vis, tvecs=tvecs, noise=noise[: T + 1].contiguous(), betas=st.betas, scene=st.scene)
        graph = m.use_hip_graph is True or (m.use_hip_graph == "auto" and desc.passes * B <= 64)
        tr = None
        with torch.cuda.device(dev):
            if graph and not any_guided and not trace:
                # hipGraph route: the loop's launches are captured once per (shape, schedule) with every pointer inside persistent
                # buffers; a call copies its inputs in, replays, and copies the results out.
                key = (B, T, int(ddim), desc.passes, desc.lbs_every_step, desc.lowprec_steps, m.gcn_precision, self._gcn_key,
                       bytes(steps), st.scene.shape[1], num_masked)
                ent = self._graphs.get(key)
                if ent is None:
                    if len(self._graphs) >= 8:
                        self._graphs.clear()
                    bufs = SimpleNamespace(**{k: torch.empty_like(v) for k, v in ins.items()}, **out_bufs())
                    bufs.mask_items, bufs.mask_slot = torch.empty_like(st.mask_items), torch.empty_like(st.mask_slot)
                    ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
                    for k, v in ins.items():
                        getattr(bufs, k).copy_(v)
                    bufs.mask_items.copy_(st.mask_items)
                    bufs.mask_slot.copy_(st.mask_slot)
                    if num_masked >= 0:      # the captured kernels read the pass map through these persistent arrays
                        _lib.check(L.ehm_gcn_set_pass_map(gcn, _lib.ptr(bufs.mask_items) if num_masked else None, _lib.ptr(bufs.mask_slot), num_masked))
                    launch(bufs, ws, None)                       # eager once: every lazy allocation inside the library happens here
                    torch.cuda.synchronize(dev)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        launch(bufs, ws, None)
                    ent = self._graphs[key] = SimpleNamespace(graph=g, bufs=bufs, ws=ws)
                for k, v in ins.items():
                    getattr(ent.bufs, k).copy_(v)
                ent.bufs.mask_items.copy_(st.mask_items)
                ent.bufs.mask_slot.copy_(st.mask_slot)
                ent.graph.replay()
                o = SimpleNamespace(**{k: getattr(ent.bufs, k).clone() for k in ("x_final", "x0", "verts", "joints", "R", "pose6d")})
            else:
                o = SimpleNamespace(**ins, **out_bufs())
                tr = torch.empty(T, B, 144, device=dev) if trace else None
                launch(o, self._workspace(nbytes, dev), tr)
        x_final, x0, verts, joints, R, pose6d = o.x_final, o.x0, o.verts, o.joints, o.R, o.pose6d
        self.last_trace = tr
        if tr is not None:
            batch["x_t"] = tr[-1]
        batch["vis_mask_smpl"] = st.vis_bool
        out = m._pack_output(batch, st, x0, pose6d, R, verts, joints)
        # a chained launch that gave up on a producer wait (GPU shared / preempted) flags the handle instead of hanging: one read-back
        # per sampling call (the call's only host wait, after everything has been enqueued) turns that into an exception rather than
        # silently wrong bodies
        # defer_status (throughput pipelines that keep batches in flight): the word is copied to pinned memory in stream order and
        # looked at by the NEXT call / by check_status(); the host does not wait here.  The flag is sticky on the device.
        with torch.cuda.device(dev):
            if defer_status:
                if getattr(self, "_status_host", None) is None:
                    self._status_host = torch.zeros(1, dtype=torch.int32).pin_memory()
                _lib.check(L.ehm_gcn_stack_status_async(gcn, self._status_host.data_ptr(), _lib.stream_ptr()), "ehm_gcn_stack_status_async")
                self._status_event = torch.cuda.Event()
                self._status_event.record()
            else:
                _lib.check(L.ehm_gcn_stack_status(gcn, _lib.stream_ptr()), "ehm_gcn_stack_status")
        return {"sample": x_final, "pred_xstart": x0, "other_outputs": out}

    def check_status(self):
        """Raise if a sampling call issued with defer_status=True flagged its chained launches (see run()).  Waits for that call."""
        ev = getattr(self, "_status_event", None)
        if ev is None:
            return
        ev.synchronize()
        self._status_event = None
        if int(self._status_host[0]) != 0:
            self._status_host.zero_()
            with torch.cuda.device(self.model.device):
                _lib.check(_lib.lib().ehm_gcn_stack_status(self.gcn(), _lib.stream_ptr()), "ehm_gcn_stack_status")
