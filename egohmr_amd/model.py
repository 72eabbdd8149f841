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

from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

from . import _lib, geometry, synthetic
from . import smpl as smpl_mod
from .encoders import ResnetPointnet, ResNet50Features
from .fused import PRECISIONS, FusedSampler  # noqa: F401  (PRECISIONS re-exported)

OPENPOSE_TO_SMPL = [8, 12, 9, 8, 13, 10, 8, 14, 11, 8, 14, 11, 0, 5, 2, 0, 5, 2, 6, 3, 7, 4, 7, 4]           # egohmr.py:111
OPENPOSE_TO_SMPL_LOOSE = [8, 13, 10, 8, 13, 10, 8, 14, 11, 8, 14, 11, 1, 5, 2, 0, 5, 2, 6, 3, 7, 4, 7, 4]     # egohmr.py:114
IMG_DIM = 2048          # columns of the GCN input feature: img 2048 | scene + transl + cam | x_t embed 512 | timestep embed 512 (EgoHMR.cond_split)
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

    # ------------------------------------------------------------------ native handle (ehm_gcn_create) - shared with FusedSampler.gcn()
    def create_native_handle(self, device):
        """ehm_gcn_create on this module's parameters -> (handle, tensors that must stay alive while it lives).  The caller destroys it."""
        import ctypes as C
        keep = []

        def params(gc, bn):
            def t(x):
                x = _lib.f32(x, device)
                keep.append(x)
                return x.data_ptr()
            p = _lib.GConvParams()
            p.W, p.M, p.adj2, p.bias = t(gc.W), t(gc.M), t(gc.adj2), t(gc.bias)
            if bn is not None:
                p.bn_weight, p.bn_bias, p.bn_mean, p.bn_var = t(bn.weight), t(bn.bias), t(bn.running_mean), t(bn.running_var)
            p.in_dim, p.out_dim = gc.in_features, gc.out_features
            return p

        gi = self.gconv_input[0]
        inp = params(gi.gconv, gi.bn)
        hidden = []
        for blk in self.gconv_layers:
            hidden += [params(blk.gconv1.gconv, blk.gconv1.bn), params(blk.gconv2.gconv, blk.gconv2.bn)]
        outp = params(self.gconv_output, None)
        arr = (_lib.GConvParams * len(hidden))(*hidden)
        adj = _lib.f32(self.adj, device)
        keep.append(adj)
        h = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.lib().ehm_gcn_create(C.byref(h), _lib.ptr(adj), C.byref(inp), arr, len(hidden), C.byref(outp), self.hid_dim,
                                                 _lib.stream_ptr()), "ehm_gcn_create")
        return h, keep

    # ------------------------------------------------------------------ the optional non-local block (modulated_gcn.py:93-94, :104-110)
    def nonlocal_packed(self):
        """The non-local block's two 1x1-conv GEMMs in ehm_conv_nhwc_split's operand format: ([theta | phi | g] weights, scale, bias),
        (W.0 with BatchNorm(eval) folded, scale, bias); re-packed when a parameter of the block changes."""
        import math
        L = _lib.lib()
        nl = self.non_local
        device = nl.theta.weight.device
        key = tuple((p.data_ptr(), p._version) for p in list(nl.parameters()) + list(nl.buffers()))
        if getattr(self, "_nl_key", None) != key:
            def pack(w2, bias):                                   # [Co, K] float32 -> X2 split weights for the conv kernel
                Co, K = w2.shape
                Co_pad = (Co + 127) // 128 * 128
                wp = torch.zeros(Co_pad, K, device=device)
                wp[:Co] = w2
                amax = float(wp.abs().max())
                scale = 2.0 ** math.floor(math.log2(2048.0 / amax)) if amax > 0 else 1.0
                buf = torch.empty(Co_pad, K, device=device)
                with torch.cuda.device(device):
                    _lib.check(L.ehm_split_pack(wp.data_ptr(), buf.data_ptr(), Co_pad, K, K, scale, _lib.stream_ptr()), "ehm_split_pack")
                return buf, scale, bias.float().contiguous()
            wqkv = torch.cat([nl.theta.weight, nl.phi.weight, nl.g.weight], 0).flatten(1).float()
            bqkv = torch.cat([nl.theta.bias, nl.phi.bias, nl.g.bias], 0)
            bn = nl.W[1]
            sc = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
            ww = (nl.W[0].weight.flatten(1).double() * sc[:, None]).float()
            bw = ((nl.W[0].bias.double() - bn.running_mean.double()) * sc + bn.bias.double()).float()
            self._nl_packed, self._nl_key = (pack(wqkv.detach(), bqkv.detach()), pack(ww.detach(), bw.detach())), key
        return self._nl_packed

    @torch.no_grad()
    def non_local_native(self, X, rows, rows_pad):
        """NONLocalBlock2D on the joint axis (modulated_gcn.py:104-110): [theta|phi|g] as ONE 1x1-conv GEMM and W + BatchNorm(eval,
        folded) + residual as another, both on ehm_conv_nhwc_split (rows = N, H = W = 1); the 24 x 24 softmax attention per body
        in ehm_nonlocal_attention.  X: float32 [rows_pad, hid] -> the same."""
        import ctypes as C
        L = _lib.lib()
        nl = self.non_local
        hid, ci = self.hid_dim, nl.inter_channels
        (wq, sq, bq), (wo, so, bo) = self.nonlocal_packed()
        s = _lib.stream_ptr()
        qkv = torch.empty(rows, 3 * ci, device=X.device)
        d = _lib.ConvDesc(X.data_ptr(), wq.data_ptr(), bq.data_ptr(), None, qkv.data_ptr(), rows, 1, 1, hid, 3 * ci, 1, 1, 1, 0, 0, sq)
        _lib.check(L.ehm_conv_nhwc_split(C.byref(d), s), "ehm_conv_nhwc_split")
        y = torch.empty(rows, ci, device=X.device)
        _lib.check(L.ehm_nonlocal_attention(qkv.data_ptr(), y.data_ptr(), rows // 24, ci, s), "ehm_nonlocal_attention")
        Z = torch.zeros(rows_pad, hid, device=X.device)
        d = _lib.ConvDesc(y.data_ptr(), wo.data_ptr(), bo.data_ptr(), X.data_ptr(), Z.data_ptr(), rows, 1, 1, ci, hid, 1, 1, 1, 0, 0, so)
        _lib.check(L.ehm_conv_nhwc_split(C.byref(d), s), "ehm_conv_nhwc_split")
        return Z

    # ------------------------------------------------------------------ ModulatedGCN.forward on its own (modulated_gcn.py:99-116)
    precision = "f16x3"      # arithmetic of the standalone call: 'f16x3' (f32-grade, default) | 'f32' | 'f16' (fused.PRECISIONS)

    def _standalone(self, device):
        """(handle, packed input-conv weights) for forward(): rebuilt when a parameter changes; its own handle (EgoHMR.fused_sampler's carries the
        sampler's pass map / precision schedule)."""
        import math
        if getattr(self, "_sa_keyfn", None) is None:
            self._sa_keyfn = _lib.TensorKey(self)
        key = (self._sa_keyfn(), str(device))
        if getattr(self, "_sa_key", None) != key:
            self._free_standalone()
            h, keep = self.create_native_handle(device)
            W = _lib.f32(self.gconv_input[0].gconv.W.detach(), device)                       # [2, in_dim, hid]
            K = W.shape[1]
            Kp = (K + 31) // 32 * 32
            Co = 2 * self.hid_dim
            Cop = (Co + 127) // 128 * 128
            w2 = torch.zeros(Cop, Kp, device=device)
            w2[:Co, :K] = W.permute(0, 2, 1).reshape(Co, K)                                   # row k * hid + n = W[k][:, n]
            amax = float(w2.abs().max())
            scale = 2.0 ** math.floor(math.log2(2048.0 / amax)) if amax > 0 else 1.0
            buf = torch.empty(Cop, Kp, device=device)
            with torch.cuda.device(device):
                _lib.check(_lib.lib().ehm_split_pack(w2.data_ptr(), buf.data_ptr(), Cop, Kp, Kp, scale, _lib.stream_ptr()), "ehm_split_pack")
            self._sa, self._sa_key = (h, keep, buf, scale, K, Kp), key
        return self._sa

    def _free_standalone(self):
        sa = getattr(self, "_sa", None)
        if sa is not None:
            try:
                _lib.lib().ehm_gcn_destroy(sa[0])
            except Exception:
                pass
            self._sa = self._sa_key = None

    def __del__(self):
        self._free_standalone()

    @torch.no_grad()
    def forward(self, x):
        """modulated_gcn.py:99-116 in eval mode on the HIP kernels: x [B, 24, in_dim] -> [B, 24, out_dim=6].

        gconv_input as a split-f16 GEMM x @ [W[0] | W[1]] (ehm_conv_nhwc_split, H = W = 1) + ehm_gcn_input_layer_rows (modulation, adjacency mix, bias,
        BatchNorm, ReLU), the residual blocks as ONE chained launch (ehm_gcn_hidden_stack), the optional non-local block, gconv_output
        (ehm_gcn_output_layer).  EgoHMR.forward / the sampler do NOT come through here: they hoist the step-invariant slices of the input feature
        (FusedSampler.prepare) - this is the module's own call surface for a user who feeds it a full feature tensor, as the reference allows."""
        import ctypes as C
        if self.training:
            raise NotImplementedError("ModulatedGCN.forward: inference only (BatchNorm in eval mode, no dropout); training is out of scope (SURVEY.md section 2)")
        if not x.is_cuda:
            raise _lib.EgoHMRHipError("ModulatedGCN.forward needs its input on a HIP device; egohmr_amd has no CPU path")
        if x.dim() != 3 or x.shape[1] != 24 or x.shape[2] != self.in_dim:
            raise ValueError(f"ModulatedGCN.forward: expected [B, 24, {self.in_dim}], got {tuple(x.shape)}")
        if self.out_dim != 6:
            raise NotImplementedError("the output-conv kernels are built for out_dim = 6 (the 6-D rotation head, egohmr.py:132)")
        from .fused import PRECISIONS
        L = _lib.lib()
        dev = x.device
        B, hid = x.shape[0], self.hid_dim
        with _lib.on_device(dev):
            h, _, wbuf, scale, K, Kp = self._standalone(dev)
            if L.ehm_gcn_get_precision(h) != PRECISIONS[self.precision]:
                _lib.check(L.ehm_gcn_set_precision(h, PRECISIONS[self.precision]), "ehm_gcn_set_precision")
            s = _lib.stream_ptr()
            rows = B * 24
            xp = torch.zeros(rows, Kp, device=dev)
            xp[:, :K] = _lib.f32(x).reshape(rows, K)
            pre = torch.empty(rows, 2 * hid, device=dev)
            d = _lib.ConvDesc(xp.data_ptr(), wbuf.data_ptr(), None, None, pre.data_ptr(), rows, 1, 1, Kp, 2 * hid, 1, 1, 1, 0, 0, scale)
            _lib.check(L.ehm_conv_nhwc_split(C.byref(d), s), "ehm_conv_nhwc_split")
            tile = L.ehm_gcn_row_tile()
            rows_pad = (rows + tile - 1) // tile * tile
            X = [torch.zeros(rows_pad, hid, device=dev) for _ in range(3)]
            _lib.check(L.ehm_gcn_input_layer_rows(h, pre.data_ptr(), X[0].data_ptr(), B, s), "ehm_gcn_input_layer_rows")
            bufs = (C.c_void_p * 3)(*[t.data_ptr() for t in X])
            res = C.c_int(0)
            _lib.check(L.ehm_gcn_hidden_stack(h, bufs, rows_pad, C.byref(res), s), "ehm_gcn_hidden_stack")
            feat = X[res.value]
            if self.nonlocal_layer:
                if self.precision == "f16":
                    raise _lib.EgoHMRHipError("the optional non-local GCN block runs on float32 features; use precision 'f16x3' or 'f32' with it")
                feat = self.non_local_native(feat, rows, rows_pad)
            out = torch.empty(B, 144, device=dev)
            _lib.check(L.ehm_gcn_output_layer(h, feat.data_ptr(), None, out.data_ptr(), B, 1, s), "ehm_gcn_output_layer")
            _lib.check(L.ehm_gcn_stack_status(h, s), "ehm_gcn_stack_status")
        return out.view(B, 24, 6)


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
        # what a clamped activation does (|x| >= 65504 in an X2 / f16 store of the denoiser: _lib.EgoHMRRangeError from the status word): 'raise', or 'f32' =
        # switch this model to gcn_precision 'f32' (float32 activations, no such limit) and run the call again.  Calls issued with defer_status=True
        # always raise (at check_status(): their results have been handed out already)
        self.on_saturation = "raise"
        # FusedSampler.run_samples: the S samples of an item run as fused loops over about this many bodies each (loop_bodies // B samples per loop, at least
        # one; 0 = all S x B bodies in one loop).  256 bodies = the three 50 MB activation matrices of the chained hidden convs stay inside the 256 MB Infinity
        # Cache and every layer is whole rounds of tiles: chain kernel 0.188 of peak against 0.174 for one 1280-body loop (config 4: 4.74 -> 5.06 k bodies/s,
        # config 3: 2.40 -> 2.50 k; profiles/r06r_loop_bodies_ab.txt).  Results do not depend on the grouping (bodies are independent).
        self.loop_bodies = 256
        self.per_step_launches = False     # True: the separate per-step launches of rounds 2-3 instead of step_fused_kernel (same bits; A/B runs and tests)
        self.pass_group = 1                # second passes pruned per item (1) or per group of this many consecutive items (FusedSampler.prepare)
        self.prune_passes = True           # exact: items whose 24 joints are all visible skip the image-masked pass (egohmr.py:239-254)
        self.overlap_encoders = False      # True: ResNet-50 and the scene PointNet on two HIP streams - measured 0.5 ms SLOWER than one after the other
                                           # (17.27 vs 16.75 ms, tools/enc_split.py, three alternations: both fill the chip on their own)
        # arithmetic of the hidden GCN convs: 'f32' (f32-input MFMA), 'f16x3' (split-f16 MFMA, f32-grade), 'f16' (plain f16, not parity-grade)
        self.gcn_precision = "f16x3"
        # arithmetic of the two conditioning encoders: 'f16x3' (split-f16, f32 grade - the parity path) | 'f16' (plain f16 operands and activations: the
        # encoders of BASELINE config 5's fp16 TIER, together with gcn_precision = 'f16'; NOT parity grade: 0.4 - 1.4 mm of final vertex, docs/EXPERIMENTS.md 3.3)
        self.encoder_precision = "f16x3"
        # precision schedule (docs/EXPERIMENTS.md 3.6): only the LAST k executed steps of a fused sampling loop run in gcn_precision ('f16x3'), the
        # earlier ones on plain f16 operands / f16 activations.  'auto' = the k that FusedSampler.calibrate_schedule MEASURED for the
        # loaded weights and the sampler in use (smallest k whose bodies stay within schedule_tol of the all-f16x3 loop; measured on the
        # first call per (weights, sampler) when auto_calibrate is on, else every step stays f16x3); an int = that k, on the caller's
        # responsibility; None = off.  There is no constant default: whether early rounding errors are contracted away depends on
        # d x0 / d x_t of the checkpoint.
        self.f16x3_last_steps = "auto"
        self.schedule_tol = 1e-5           # metres, max vertex / joint distance to the all-f16x3 loop (the north-star bar is 1e-4)
        self.auto_calibrate = True
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
        with _lib.on_device(self.device):                 # native calls launch on the CURRENT device's stream
            return self._forward_on_device(batch, timesteps, eval_with_uncond)

    def _forward_on_device(self, batch, timesteps, eval_with_uncond):
        fs = self.fused_sampler
        st = fs.prepare(batch)
        x_t = _lib.f32(batch["x_t"], self.device).reshape(-1, 144)
        B = x_t.shape[0]
        passes = 2 if (self.diffuse_fuse and eval_with_uncond) else 1
        # egohmr.py:178 embeds `timesteps` [bs] row by row.  The samplers always pass one value for the whole batch (gaussian_diffusion.py:495):
        # that is the one-launch route; a direct call with different values per item runs one denoiser evaluation per distinct value.
        ts = torch.as_tensor(timesteps, device=self.device).reshape(-1).long()
        if ts.numel() not in (1, B):
            raise ValueError(f"timesteps must hold one value per item: got {ts.numel()} for a batch of {B}")
        uniq = torch.unique(ts) if ts.numel() > 1 else ts[:1]
        if uniq.numel() == 1:
            x0 = fs.denoise_once(st, x_t, fs.timestep_vectors(uniq)[0], passes)
        else:
            x0 = torch.empty(B, 144, device=self.device)
            tv = fs.timestep_vectors(uniq)
            for i, t in enumerate(uniq.tolist()):
                sel = torch.nonzero(ts == t).reshape(-1)
                x0[sel] = fs.denoise_once(fs._subset(st, sel), x_t[sel].contiguous(), tv[i], passes)
        mean, std = self._std_mean()
        verts = torch.empty(B, self.smpl.num_verts, 3, device=self.device)
        joints = torch.empty(B, self.smpl.num_joints_out, 3, device=self.device)
        R = torch.empty(B, 24, 3, 3, device=self.device)
        pose6d = torch.empty(B, 144, device=self.device)
        _lib.check(_lib.lib().ehm_smpl_forward_rot6d(self.smpl.handle(), _lib.ptr(st.betas), _lib.ptr(x0), _lib.ptr(mean), _lib.ptr(std),
                                                     _lib.ptr(verts), _lib.ptr(joints), _lib.ptr(R), _lib.ptr(pose6d), None, B,
                                                     _lib.stream_ptr()), "ehm_smpl_forward_rot6d")
        batch["vis_mask_smpl"] = st.vis_bool
        return self._pack_output(batch, st, x0, pose6d, R, verts, joints, chk=x_t.contiguous(), chk_rows=1)

    def _pack_output(self, batch, st, x0, pose6d, R, verts, joints, chk=None, chk_rows=0, last_noise=None, x_final=None):
        """The output dict of EgoHMR.forward (egohmr.py:283-303) in one launch (ehm_pack_outputs): items with a NaN / Inf in their inputs
        (st.finite) or in a row of `chk` come out as NaN, like the reference's float32 graph gives them."""
        import ctypes as C
        B, J, dev = x0.shape[0], joints.shape[1], x0.device
        buf = torch.empty(B * (10 + 216 + 5 * J + 4), device=dev)
        cuts, off = [], 0
        for n in (10, 9, 207, 3 * J, 2 * J, 2, 2):
            cuts.append(buf[off:off + B * n].view(B, n))
            off += B * n
        betas, go, bp, kp3d, kp2d, focal, center = cuts
        d = _lib.PackDesc(B=B, J=J, V=verts.shape[1], finite=st.finite.data_ptr(), chk=chk.data_ptr() if chk is not None else None, chk_rows=int(chk_rows),
                          last_noise=last_noise.data_ptr() if last_noise is not None else None, x_final=x_final.data_ptr() if x_final is not None else None,
                          x0=x0.data_ptr(), pose6d=pose6d.data_ptr(), R=R.data_ptr(), verts=verts.data_ptr(), joints=joints.data_ptr(),
                          betas_in=st.betas.data_ptr(), betas_out=betas.data_ptr(), transl=st.transl.data_ptr(), fx=st.fx.data_ptr(), cx=st.cam_cx.data_ptr(),
                          cy=st.cam_cy.data_ptr(), fx_norm=self.cfg.CAM.FX_NORM_COEFF, global_orient=go.data_ptr(), body_pose=bp.data_ptr(),
                          kp3d_full=kp3d.data_ptr(), kp2d_full=kp2d.data_ptr(), focal=focal.data_ptr(), center=center.data_ptr(), finite_out=None)
        for t in (x0, pose6d, R, verts, joints, st.betas, st.transl, st.fx, st.cam_cx, st.cam_cy, st.finite):
            assert t.is_contiguous()
        with _lib.on_device(dev):
            _lib.check(_lib.lib().ehm_pack_outputs(C.byref(d), _lib.stream_ptr()), "ehm_pack_outputs")
        self.scene_pcd_verts = st.scene
        self.input_transl = st.transl
        self.smpl_output = smpl_mod.SMPLOutput(vertices=verts, joints=joints, full_pose=R)
        self.focal_length, self.camera_center_full = focal, center                     # :283-285
        return {
            "pred_x_start": x0,
            "pred_smpl_params": {"global_orient": go.view(B, 1, 3, 3), "body_pose": bp.view(B, 23, 3, 3), "betas": betas},
            "pred_pose_6d": pose6d,
            "pred_keypoints_3d": joints,
            "pred_vertices": verts,
            "pred_keypoints_3d_full": kp3d.view(B, J, 3),
            "pred_keypoints_2d_full": kp2d.view(B, J, 2),                              # :295-301
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
        # bbox-selected points in BOTH reference files (egohmr.py:499-505, egohmr_volsmpl.py:531-537), whatever the guidance uses
        _, _, hits = self.fused_sampler.collision(so.vertices, self.scene_pcd_verts, want_grad=False, want_hits=True, all_points=False)
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
    VolumetricSMPL is a learned SDF that cannot be obtained offline: the collision term is the build's proxy (docs/EXPERIMENTS.md 3.5),
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
