"""The native sampling engine behind `EgoHMR.fused_sampler`: step-invariant conditioning (`prepare`), the one-call sampling loop
(`run` / `run_samples` -> ehm_sample_loop), the granular denoiser evaluation of `EgoHMR.forward`, the collision-guidance pieces and
the per-checkpoint precision-schedule calibration (`calibrate_schedule`).

Reference code this replaces on the hot path: diffusion/gaussian_diffusion.py:391-508 / :618-718 (the loops) driving
models/egohmr/egohmr.py:173-303 (forward) and :517-570 (guide_coll); see DESIGN.md sections 2 and 4.
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

import torch

from . import _lib
from . import smpl as smpl_mod

PRECISIONS = {"f32": 0, "f16x3": 1, "f16": 2}

# ---------------------------------------------------------------------------------------------- native engine
class _Prepared(SimpleNamespace):
    pass


class FusedSampler:
    """Owns the native denoiser handle and runs sampling loops through ehm_sample_loop."""

    def __init__(self, model):
        self._model_ref = [model]
        self._gcn = None
        self._gcn_key = None
        self._folded = None
        self._prep_key = None
        self._prep = None
        self._ws = None
        self._graphs = {}
        self._sched_cache = {}          # schedule_key -> calibration info (calibrate_schedule)
        self.schedule_info = None       # the calibration the most recent 'auto' run used (None: ran all-f16x3 / explicit k)
        self.last_lowprec = 0
        self.last_trace = None

    @property
    def model(self):
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
            gi = dm.gconv_input[0]
            h, self._gcn_keep = dm.create_native_handle(m.device)             # (ModulatedGCN owns the parameter marshalling: ehm_gcn_create)
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
        if self.model.diffusion_model.nonlocal_layer:       # the one-call loop runs the block natively (ehm_gcn_set_nonlocal)
            (wq, sq, bq), (wo, so, bo) = self._nonlocal_packed()
            if getattr(self, "_nl_set", None) != (self._nl_key, self._gcn.value):
                p = _lib.NonlocalParams(wq.data_ptr(), bq.data_ptr(), sq, wo.data_ptr(), bo.data_ptr(), so, self.model.diffusion_model.non_local.inter_channels)
                _lib.check(_lib.lib().ehm_gcn_set_nonlocal(self._gcn, C.byref(p)), "ehm_gcn_set_nonlocal")
                self._nl_set = (self._nl_key, self._gcn.value)
        return self._gcn

    def _free(self):
        if self._gcn is not None:
            try:
                _lib.lib().ehm_gcn_destroy(self._gcn)
            except Exception:
                pass
            self._gcn = None
            self._nl_set = None          # a new handle may reuse the freed one's address: the non-local block must be installed again

    def __del__(self):
        self._free()

    def _backbone_fn(self):
        """ResNet-50 with BatchNorm folded into the convolutions, rebuilt when the backbone weights change."""
        return self.model.backbone.current()

    # ------------------------------------------------------------------ step-invariant conditioning
    @torch.no_grad()
    def prepare(self, batch) -> _Prepared:
        """Everything in EgoHMR.forward that does not depend on x_t / t (egohmr.py:182-223, :263-265)."""
        with _lib.on_device(self.model.device):          # every native call below launches on the CURRENT device's stream
            return self._prepare_on_device(batch)

    def _prepare_on_device(self, batch) -> _Prepared:
        m, L = self.model, _lib.lib()
        # Cache key = identity AND version of every tensor the conditioning is computed from, and of every weight it passes
        # through.  The cached entry keeps strong references to those input tensors, so neither their id() nor their storage can be
        # recycled for a different batch while the entry is alive; in-place edits bump _version.
        ins = [batch["img"], batch["scene_pcd_verts_full"], batch["orig_keypoints_2d"], batch["fx"], batch["cam_cx"], batch["cam_cy"],
               batch["box_center"], batch["box_size"], batch["smpl_params"]["transl"]]
        if ins[0].shape[0] == 0:
            raise ValueError("empty batch (the reference fails on it too: egohmr.py:233 reshapes x_t [0, 144] to [0, 24, -1])")
        # ... and of the model switches that ehm_item_prep bakes into the cached state (the pass map's grouping, the camera-feature columns, the
        # scene frame, which OpenPose joints feed the visibility mask)
        if m.encoder_precision not in ("f16x3", "f16"):
            raise ValueError(f"EgoHMR.encoder_precision must be 'f16x3' or 'f16', not {m.encoder_precision!r}")
        m.backbone.hi_only = m.scene_enc.hi_only = m.encoder_precision == "f16"
        switches = (int(getattr(m, "pass_group", 1)), bool(m.with_bbox_info), bool(m.with_cam_center), bool(m.scene_cano), tuple(m.openpose_to_smpl),
                    m.encoder_precision)
        key = tuple((id(t), t._version, t.data_ptr()) for t in ins) + self._param_key() + self._cond_param_key() + (switches,)
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
        B, f, st = img.shape[0], self._folded, _lib.stream_ptr()
        # ---- per-item scalars in two launches (csrc/prep.hip): joint visibility (:186-188), the pass-pruning map (ehm_gcn_set_pass_map: items
        # with an invisible joint need the second pass), TranslEnc (:217) and the camera features (:195-205) written straight into the padded
        # [scene | transl | cam] operand of the projections, and the per-item "inputs are finite" flag.  The reference's float32 graph carries a
        # NaN / Inf of an item's inputs into every output of THAT item (ReLU and max-pool propagate NaN in torch); the kernels' v_max /
        # saturating conversions would swallow it, so the flag travels beside the data (ehm_pack_outputs).  A row sum is non-finite exactly
        # when the row holds a NaN / Inf - or finite values whose float32 sum overflows, which no image or point cloud in metres does.
        # The count of second passes is the ONE host read-back of a batch.  It is REQUESTED here (an asynchronous copy into pinned memory +
        # an event) and LOOKED AT after the encoders have been enqueued: in a pipeline of batches the copy sits behind the previous batch's
        # sampling loop, the host spends that time enqueuing this batch's encoders, and the GPU never waits for Python.
        kp = g("orig_keypoints_2d")
        fx, cx, cy, bc, bs = g("fx"), g("cam_cx"), g("cam_cy"), g("box_center"), g("box_size")
        te = m.transl_enc.layers
        tw = [_lib.f32(te[0].weight, dev), _lib.f32(te[0].bias, dev), _lib.f32(te[2].weight, dev), _lib.f32(te[2].bias, dev)]
        n_scene, n_tr = m.scene_enc.fc_c.out_features, te[2].out_features
        n_other = n_scene + n_tr + 1 + (3 if m.with_bbox_info else 0) + (2 if m.with_cam_center else 0)
        jm = getattr(self, "_joint_map", None)
        if kp.dim() != 3 or kp.shape[2] != 3 or not all(0 <= int(k) < kp.shape[1] for k in m.openpose_to_smpl):
            # (item_prep_kernel indexes keypoints_2d[b, joint_map[t], 2]: checked here, once per batch, instead of on the device)
            raise ValueError(f"orig_keypoints_2d must be [B, NK, 3] with NK > max(openpose_to_smpl) = {max(m.openpose_to_smpl)}; got {tuple(kp.shape)}")
        if jm is None or jm[0] != (tuple(m.openpose_to_smpl), str(dev)):
            jm = self._joint_map = ((tuple(m.openpose_to_smpl), str(dev)), torch.tensor(m.openpose_to_smpl, dtype=torch.int32, device=dev))
        oth = torch.empty(B, f.k_oth, device=dev)
        vis = torch.empty(B, 24, dtype=torch.uint8, device=dev)
        flags = torch.empty(2, B, dtype=torch.uint8, device=dev)                       # finite | need (scratch)
        maps = torch.empty(2 * B + 1, dtype=torch.int32, device=dev)                   # mask_slot | mask_items | count
        sums = (img.reshape(B, -1).sum(dim=1), scene.reshape(B, -1).sum(dim=1))
        d = _lib.ItemPrepDesc(keypoints_2d=kp.data_ptr(), joint_map=jm[1].data_ptr(), NK=kp.shape[1], force_visible=8, fx=fx.data_ptr(), cx=cx.data_ptr(),
                              cy=cy.data_ptr(), box_center=bc.data_ptr(), box_size=bs.data_ptr(), transl=transl.data_ptr(), fx_norm=m.cfg.CAM.FX_NORM_COEFF,
                              with_bbox=int(m.with_bbox_info), with_cam_center=int(m.with_cam_center), tW1=tw[0].data_ptr(), tb1=tw[1].data_ptr(),
                              tW2=tw[2].data_ptr(), tb2=tw[3].data_ptr(), t_hidden=te[0].out_features, t_out=n_tr, img_rowsum=sums[0].data_ptr(),
                              scene_rowsum=sums[1].data_ptr(), other=oth.data_ptr(), other_ld=f.k_oth, other_col0=n_scene, vis=vis.data_ptr(),
                              mask_slot=maps.data_ptr(), mask_items=maps[B:].data_ptr(), count=maps[2 * B:].data_ptr(), finite=flags.data_ptr(),
                              need_scratch=flags[1].data_ptr(), pass_group=int(getattr(m, "pass_group", 1)), B=B)
        _lib.check(L.ehm_item_prep(C.byref(d), st), "ehm_item_prep")
        if getattr(self, "_count_host", None) is None:
            self._count_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._count_host.copy_(maps[2 * B:], non_blocking=True)
        count_ready = torch.cuda.Event()
        count_ready.record(torch.cuda.current_stream(dev))
        # The two encoders are independent, but each fills the chip on its own: two HIP streams measured slower than one (model.overlap_encoders).
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
        oth[:, :n_scene].copy_(scene_feats)                                            # :220-221 [scene | transl | cam] (the rest was written by ehm_item_prep)
        h_img, h_oth, betas = self._project(img_feats.contiguous(), oth, n_other)
        # items with a non-finite input (flags[0] == 0: their outputs are NaN by the packer's rule) get ZERO conditioning: the encoders' saturating f16 stores
        # have turned their inf / NaN into large FINITE features, which would drive the denoiser's activations to the f16 range and raise its range guard
        # (status bit 2) for an item that is already accounted for
        bad = (flags[0] == 0).view(B, 1, 1)
        h_img.masked_fill_(bad, 0.0)
        h_oth.masked_fill_(bad, 0.0)
        count_ready.synchronize()
        num_masked = int(self._count_host[0])
        mask_slot, mask_items = maps[:B], maps[B:B + num_masked]
        self._prep = _Prepared(B=B, h_img=h_img, h_oth=h_oth, vis=vis, vis_bool=vis.view(torch.bool),
                               betas=betas, scene=scene, transl=transl, fx=fx, cam_cx=cx, cam_cy=cy, img_feats=img_feats,
                               scene_feats=scene_feats, finite=flags[0].view(torch.bool))
        self._prep.inputs = ins                  # strong references (see the key above)
        self._prep.mask_items, self._prep.mask_slot, self._prep.num_masked = mask_items, mask_slot, num_masked
        self._prep_key = key
        return self._prep

    def _project(self, img_feats, oth, n_other):
        """The step-invariant slices of the input graph conv ([B,2,hid] each: image features, scene + translation + camera features)
        and the beta head (egohmr.py:263-265, Linear -> ReLU -> Linear + init_betas) as exact-float32 matrix-core GEMMs built for M = B
        rows (ehm_skinny_gemm_f32).  oth = [scene | transl | cam] zero-padded to a multiple of 32 columns, n_other of them live."""
        m, f, L = self.model, self._folded, _lib.lib()
        B, dev, hid = img_feats.shape[0], img_feats.device, m.diffusion_model.hid_dim
        st = _lib.stream_ptr()
        a = img_feats.shape[1]
        l1, l2 = m.beta_layer.layers[0], m.beta_layer.layers[2]
        if a % 32 or l1.out_features % 32 or l1.in_features != a + n_other or l2.out_features > 32:
            raise _lib.EgoHMRHipError(f"conditioning widths the projection kernels are not built for: image features {a}, beta head "
                                      f"{l1.in_features} -> {l1.out_features} -> {l2.out_features}, other features {n_other}")
        h_img = torch.empty(B, 2, hid, device=dev)
        h_oth = torch.empty(B, 2, hid, device=dev)
        _lib.check(L.ehm_skinny_gemm_f32(_lib.ptr(img_feats), _lib.ptr(f.W_img_cat), None, _lib.ptr(h_img), B, a, 2 * hid, 0, st), "ehm_skinny_gemm_f32")
        _lib.check(L.ehm_skinny_gemm_f32(_lib.ptr(oth), _lib.ptr(f.W_oth_cat), None, _lib.ptr(h_oth), B, f.k_oth, 2 * hid, 0, st), "ehm_skinny_gemm_f32")
        # beta head: both layers on the same kernel (weights transposed and zero-padded once per weight version; init_betas folded into
        # the second bias)
        ib = m.beta_layer.init_betas
        key = tuple((t.data_ptr(), t._version) for t in (l1.weight, l1.bias, l2.weight, l2.bias, ib)) + (str(dev),)
        if getattr(self, "_beta_key", None) != key:
            Wt = torch.zeros(a + f.k_oth, l1.out_features, device=dev)
            w = l1.weight.detach().float().to(dev)
            Wt[:a] = w[:, :a].t()
            Wt[a:a + n_other] = w[:, a:].t()
            W2 = torch.zeros(l1.out_features, 32, device=dev)
            W2[:, :l2.out_features] = l2.weight.detach().float().to(dev).t()
            b2 = torch.zeros(32, device=dev)
            b2[:l2.out_features] = (l2.bias.detach().float().to(dev) + ib.detach().float().to(dev).reshape(-1))
            self._beta_w = (Wt.contiguous(), l1.bias.detach().float().to(dev).contiguous(), W2.contiguous(), b2)
            self._beta_key = key
        W1, b1, W2, b2 = self._beta_w
        xb = torch.cat([img_feats, oth], dim=1)
        hb = torch.empty(B, l1.out_features, device=dev)
        _lib.check(L.ehm_skinny_gemm_f32(_lib.ptr(xb), _lib.ptr(W1), _lib.ptr(b1), _lib.ptr(hb), B, xb.shape[1], l1.out_features, 1, st), "ehm_skinny_gemm_f32")
        b32 = torch.empty(B, 32, device=dev)
        _lib.check(L.ehm_skinny_gemm_f32(_lib.ptr(hb), _lib.ptr(W2), _lib.ptr(b2), _lib.ptr(b32), B, l1.out_features, 32, 0, st), "ehm_skinny_gemm_f32")
        return h_img, h_oth, b32[:, :l2.out_features].contiguous()

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
    def timestep_vectors(self, t_orig) -> torch.Tensor:
        """[n] original timesteps -> [n,2,hid]: TimestepEmbedder (egohmr.py:642-643) pushed through the timestep
        slice of the input conv, plus the folded InputProcess bias.  A python sequence of ints (the sampler's timestep map) is cached per
        (weights, sequence): the embedding MLP and its float64 projection ran again on every sampling call."""
        m = self.model
        self.gcn()
        ckey = None
        if not torch.is_tensor(t_orig):
            ckey = (self._param_key(), self._cond_param_key(), tuple(int(t) for t in t_orig))
            hit = getattr(self, "_tv_cache", None)
            if hit is not None and hit[0] == ckey:
                return hit[1]
            t_orig = torch.tensor(ckey[2], device=m.device, dtype=torch.long)
        temb = m.embed_timestep.time_embed(m.sequence_pos_encoder.pe[t_orig][:, 0])    # [n,512]
        tv = torch.einsum("ne,kef->nkf", temb.double(), self._folded.W_t) + self._folded.bx[None]
        tv = tv.float().contiguous()
        if ckey is not None:
            self._tv_cache = (ckey, tv)
        return tv

    @staticmethod
    def step_table(diffusion, ddim, cond_grad_weight, guided):
        """The T ehm_step_coefs rows of a loop (newest first = execution order), cached on the diffusion object: every row costs a handful of
        float32 CPU tensor ops (GaussianDiffusion.step_coefs keeps torch's roundings), 100 rows ~ 3 ms of host time per call."""
        cache = diffusion.__dict__.setdefault("_ehm_step_tables", {})
        # (the fingerprint of the tables the rows are computed from: a diffusion object whose betas / timestep map were edited in place gets new rows)
        fp = hash((diffusion.betas.tobytes(), tuple(getattr(diffusion, "timestep_map", ()))))
        key = (bool(ddim), float(cond_grad_weight), bool(guided), fp)
        if key not in cache:
            T = diffusion.num_timesteps
            rows = [diffusion.step_coefs(i, ddim, 0.0, cond_grad_weight, guided) for i in range(T - 1, -1, -1)]
            first_guided = next((i for i, r in enumerate(rows) if r.grad_scale != 0.0), T)
            cache[key] = ((_lib.StepCoefs * T)(*rows), first_guided)
        return cache[key]

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

    def _non_local(self, X, rows, rows_pad):
        """NONLocalBlock2D on the joint axis (modulated_gcn.py:104-110): ModulatedGCN.non_local_native."""
        return self.model.diffusion_model.non_local_native(X, rows, rows_pad)

    def _nonlocal_packed(self):
        dm = self.model.diffusion_model
        self._nl_packed = dm.nonlocal_packed()
        self._nl_key = dm._nl_key
        return self._nl_packed

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

    # ------------------------------------------------------------------ precision schedule: calibrated per checkpoint
    def schedule_key(self, diffusion, ddim: bool, n_guided: int, cond_grad_weight: float = 0.0, guide_denom: float = 1.0):
        """What a calibrated k is valid for: these denoiser / embedder weights (identity + version of every tensor), this sampler
        (original timesteps visited, ancestral or DDIM), this guidance window, weight and reduction, this tolerance.  The batch size behind a
        `-loss.mean()` (guide_denom) is NOT part of the key: a ragged last batch, or ranks with different shard sizes, must find the job's one
        calibration (the f16 steps end >= 8 steps before the guided window whatever its strength)."""
        m = self.model
        return (self._param_key(), self._cond_param_key(), tuple(diffusion.timestep_map), int(getattr(diffusion, "original_num_steps", diffusion.num_timesteps)),
                bool(ddim), int(n_guided), (float(cond_grad_weight), str(m.guide_reduction)) if n_guided else None, bool(m.guide_all_points) if n_guided else False,
                bool(m.diffuse_fuse),
                float(m.schedule_tol))

    def lowprec_steps(self, T: int, guided=False, ddim: bool = True, key=None) -> int:
        """How many LEADING steps of a T-step fused loop run on plain f16 operands (EgoHMR.f16x3_last_steps): None -> 0 (every step
        f32-grade), an int k -> T - k (the caller vouches for it), 'auto' -> T - k for the k that `calibrate_schedule` measured for
        `key` (= schedule_key(...)) on the LOADED weights, and 0 when there is no calibration for it.  There is no constant policy
        any more: a k tuned on one network says nothing about another (docs/EXPERIMENTS.md 3.6)."""
        k = self.model.f16x3_last_steps
        if k is None or self.model.gcn_precision != "f16x3":
            return 0
        if k == "auto":
            info = self._sched_cache.get(key) if key is not None else None
            if info is None:
                return 0
            k = info["k"]
        return max(0, T - int(k))

    @staticmethod
    def _k_ladder(T: int, floor: int = 2):
        """Candidate values of k (last k steps in f16x3), ascending, roughly geometric (ratio 4/3), always ending at T."""
        ks, x = [], float(max(floor, 2))
        while x < T:
            if not ks or int(round(x)) > ks[-1]:
                ks.append(int(round(x)))
            x *= 4.0 / 3.0
        return [k for k in ks if k < T] + [T]

    @staticmethod
    def pick_k(ladder, err_a, err_b, bar):
        """Index into `ladder` (ascending k, last entry = T = every step f32-grade, error 0 by definition) of the smallest k whose error
        on draw A is <= bar AND, from there upwards, the first k that also passes on draw B.  Bisection on draw A (the error falls with
        k up to noise; a non-monotone blip can only make the answer more conservative because draw B re-checks it), then a linear climb."""
        lo, hi = 0, len(ladder) - 1
        while lo < hi:
            mid = (lo + hi) // 2
            if err_a(ladder[mid]) <= bar:
                hi = mid
            else:
                lo = mid + 1
        idx = lo
        while idx < len(ladder) - 1 and err_b(ladder[idx]) > bar:
            idx += 1
        return idx

    def _subset(self, st, sel):
        """Some items of a prepared batch as a prepared batch of their own (pass map recomputed): sel = n (the first n) or an index tensor."""
        if isinstance(sel, int):
            if sel >= st.B:
                return st
            sel = torch.arange(sel, device=st.vis.device)
        sel = sel.to(st.vis.device, torch.long)
        fields = {k: (v.index_select(0, sel).contiguous() if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == st.B and k not in ("mask_items", "mask_slot") else v)
                  for k, v in vars(st).items()}
        r = _Prepared(**fields)
        r.B = int(sel.numel())
        need = ~r.vis_bool.all(dim=1)
        order = torch.argsort((~need).to(torch.uint8), stable=True).to(torch.int32)
        r.mask_slot = torch.where(need, torch.cumsum(need.to(torch.int32), 0) - 1, torch.full_like(need, -1, dtype=torch.int32)).to(torch.int32).contiguous()
        r.num_masked = int(need.sum())
        r.mask_items = order[: r.num_masked].contiguous()
        return r

    @torch.no_grad()
    def calibrate_schedule(self, diffusion, batch=None, ddim=False, guided=False, cond_grad_weight=1.0, tol=None, bodies=64, prepared=None,
                           seeds=(20260929, 20260930), force=False, denom_items=None, n_guided=None):
        """Measure, for the weights that are loaded NOW, the smallest k such that a sampling loop whose first T - k steps run the hidden
        convs on plain f16 operands ends within `tol` metres (max vertex / joint distance, every body) of the loop that runs every step in
        split-f16 (f32-grade) arithmetic on the same noise - and cache it under `schedule_key`.

        Why per checkpoint: x_{t-1} = c1 x0(x_t) + c2 x_t carries an early step's rounding error with gain c1 J + c2, J = d x0 / d x_t.
        A denoiser that ignores x_t (J ~ 0) contracts it away within a few steps; a trained START_X denoiser has J -> 1 / sqrt(abar_t)
        at low noise, where c1 J + c2 = 1 / sqrt(alpha_t) >= 1: the error is carried to the output (gaussian_diffusion.py:298-337 is exact
        for any weights, so must this be).  Procedure: the first `bodies` items of the batch (conditioning already encoded), two private noise
        draws; bisection over a geometric ladder of k with draw A against tol / 2, then draw B must pass too (k moves up the ladder until
        it does).  k = T (no f16 step at all) always passes, so the result is always safe; cost = ~6-10 small sampling loops, once per
        (weights, sampler).  Returns the info dict that `schedule_info` / bench.py report."""
        m = self.model
        if m.gcn_precision != "f16x3":
            raise _lib.EgoHMRHipError("calibrate_schedule: the precision schedule only exists for gcn_precision='f16x3'")
        tol = float(m.schedule_tol if tol is None else tol)
        st_full = prepared if prepared is not None else self.prepare(batch)
        T = diffusion.num_timesteps
        if n_guided is None:
            n_guided = sum(1 for i in range(min(T, 16)) if diffusion.step_coefs(i, ddim, 0.0, cond_grad_weight, guided).grad_scale != 0.0)   # guided tail: t <= 10
        denom_items = int(denom_items or st_full.B)
        old_tol, m.schedule_tol = m.schedule_tol, tol
        try:
            key = self.schedule_key(diffusion, ddim, n_guided, cond_grad_weight, self.guide_denom(denom_items))
        finally:
            m.schedule_tol = old_tol
        if not force and key in self._sched_cache:
            return self._sched_cache[key]
        # always `bodies` bodies: the finite items of the batch, replicated by index when there are fewer (each copy gets its own noise), so
        # that a first call with B = 2 does not fix k from two bodies for every later batch; items with non-finite inputs are left out (they
        # would turn every trial distance into NaN and cache k = T)
        good = torch.nonzero(st_full.finite).reshape(-1)
        if good.numel() == 0:
            return {"k": int(T), "T": int(T), "f16_steps": 0, "tol_m": tol, "criterion": "no item with finite inputs in the batch: not calibrated, not cached",
                    "bodies": 0, "ddim": bool(ddim), "guided_steps": int(n_guided), "trials": []}
        st = self._subset(st_full, good[torch.arange(int(bodies), device=good.device) % good.numel()])
        nb, dev = st.B, m.device
        sub_batch = dict(batch) if batch is not None else {}

        def loop(noise, lowprec):
            r = self.run(diffusion, sub_batch, noise, ddim=ddim, guided=guided, cond_grad_weight=cond_grad_weight, prepared=st,
                         denom_items=denom_items, lowprec=lowprec)
            o = r["other_outputs"]
            return o["pred_vertices"].clone(), o["pred_keypoints_3d"].clone()

        def dist(a, b):
            return max(float((a[0] - b[0]).norm(dim=-1).max()), float((a[1] - b[1]).norm(dim=-1).max()))

        draws, refs, tried = [], [], {}
        for sd_ in seeds:
            g = torch.Generator(device=dev).manual_seed(int(sd_))
            draws.append(torch.randn(T + 1, nb, 144, device=dev, generator=g))
        refs.append(loop(draws[0], 0))
        # the f16 steps never reach into the guided window: the guidance feeds nearest-vertex switches back with gain
        ladder = self._k_ladder(T, floor=(n_guided + 8) if n_guided else 2)

        def err(k, d):
            if (k, d) not in tried:
                tried[(k, d)] = 0.0 if k >= T else dist(loop(draws[d], T - k), refs[d])
            return tried[(k, d)]

        def err1(k):
            if ladder[-1] > k and len(refs) < 2:          # draw B's reference loop only when a candidate below T reaches the second check
                refs.append(loop(draws[1], 0))
            return err(k, 1)

        idx = self.pick_k(ladder, lambda k: err(k, 0), err1, 0.5 * tol)
        k = ladder[idx]
        info = {"k": int(k), "T": int(T), "f16_steps": int(T - k), "tol_m": tol, "criterion": "max vertex/joint distance to the all-f16x3 loop <= tol/2 on two noise draws",
                "bodies": int(nb), "ddim": bool(ddim), "guided_steps": int(n_guided),
                "trials": sorted([{"k": kk, "draw": d, "max_dist_m": e} for (kk, d), e in tried.items()], key=lambda r: (r["k"], r["draw"]))}
        self._sched_cache[key] = info
        self.schedule_info = info
        return info

    def install_schedule(self, diffusion, info, ddim=False, guided=False, cond_grad_weight=1.0, denom_items=1):
        """Adopt a calibration result measured elsewhere (another rank's: egohmr_amd.dist.agree_schedule) for THIS process's weights."""
        T = diffusion.num_timesteps
        n_guided = sum(1 for i in range(min(T, 16)) if diffusion.step_coefs(i, ddim, 0.0, cond_grad_weight, guided).grad_scale != 0.0)
        assert int(info["T"]) == T, (info["T"], T)
        self._sched_cache[self.schedule_key(diffusion, ddim, n_guided, cond_grad_weight, self.guide_denom(int(denom_items)))] = dict(info)
        self.schedule_info = dict(info)

    @torch.no_grad()
    def measure_gain(self, batch=None, timesteps=(0,), prepared=None, bodies=16, delta=1e-2, seed=7):
        """Directional sensitivity of the loaded denoiser, || x0(x_t + d) - x0(x_t) || / || d || for a random direction d at x_t ~ N(0, 1),
        averaged over `bodies` items, per ORIGINAL timestep - the J that decides whether early rounding errors are contracted
        (calibrate_schedule).  Evaluated through the product's own denoiser in its f32-grade arithmetic."""
        m = self.model
        st = self._subset(prepared if prepared is not None else self.prepare(batch), bodies)
        g = torch.Generator(device=m.device).manual_seed(seed)
        x = torch.randn(st.B, 144, device=m.device, generator=g)
        d = torch.randn(st.B, 144, device=m.device, generator=g)
        d = d / d.norm(dim=1, keepdim=True) * delta
        passes = 2 if m.diffuse_fuse else 1
        out = {}
        for t in timesteps:
            tv = self.timestep_vectors(torch.tensor([int(t)], device=m.device))[0]
            a = self.denoise_once(st, x, tv, passes).clone()
            b = self.denoise_once(st, (x + d).contiguous(), tv, passes)
            out[int(t)] = float(((b - a).norm(dim=1) / delta).mean())
        return out

    # ------------------------------------------------------------------ S samples of one batch in ONE loop
    @torch.no_grad()
    def run_samples(self, diffusion, batch, noise_stacks, ddim=False, guided=False, cond_grad_weight=1.0, defer_status=False):
        """The reference draws S samples per item with S sequential sampling loops over the same batch (test_egohmr.py:251-266).  The
        samples are independent given the conditioning, so this runs them as loops over g*B bodies (sample-major: body s*B + b) with
        the conditioning replicated by index - the same arithmetic per body (the guidance denominator stays B), fewer launches and
        full-size conv tiles for small B.  g = EgoHMR.loop_bodies // B samples share a loop (at least one): about 256 bodies per loop keep the
        three activation matrices of the chained hidden convs (50 MB each) inside the 256 MB Infinity Cache and give every layer whole rounds
        of tiles - ONE loop over 1280 bodies ran the chain kernel at 0.173 of peak, loops of 256 at 0.188 (profiles/r06r_loop_bodies_ab.txt).
        noise_stacks: S tensors [T+1,B,144].  Returns a list of S result dicts like run()."""
        S = len(noise_stacks)
        st = self.prepare(batch)
        if S == 1:
            return [self.run(diffusion, batch, noise_stacks[0], ddim=ddim, guided=guided, cond_grad_weight=cond_grad_weight, prepared=st,
                             defer_status=defer_status)]
        B = st.B
        width = int(getattr(self.model, "loop_bodies", 0) or 0)
        g = S if width <= 0 else max(1, min(S, width // max(B, 1)))
        outs = []
        for s0 in range(0, S, g):
            outs += self._run_sample_group(diffusion, batch, st, noise_stacks[s0:s0 + g], ddim, guided, cond_grad_weight, defer_status)
        # leave the model's per-call attributes as S sequential calls would: un-replicated inputs, the last sample's bodies
        m = self.model
        m.scene_pcd_verts, m.input_transl = st.scene, st.transl
        m.focal_length, m.camera_center_full = m.focal_length[:B], m.camera_center_full[:B]
        last = outs[-1]["other_outputs"]
        m.smpl_output = smpl_mod.SMPLOutput(vertices=last["pred_vertices"], joints=last["pred_keypoints_3d"],
                                            full_pose=torch.cat([last["pred_smpl_params"]["global_orient"], last["pred_smpl_params"]["body_pose"]], dim=1))
        return outs

    def _run_sample_group(self, diffusion, batch, st, noise_stacks, ddim, guided, cond_grad_weight, defer_status):
        """S samples of the prepared batch `st` as ONE loop over S*B bodies -> S result dicts."""
        S, B = len(noise_stacks), st.B
        if S == 1:
            return [self.run(diffusion, dict(batch), noise_stacks[0], ddim=ddim, guided=guided, cond_grad_weight=cond_grad_weight, prepared=st, denom_items=B,
                             defer_status=defer_status)]
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
        return split(res)

    # ------------------------------------------------------------------ whole loop
    @torch.no_grad()
    def run(self, diffusion, batch, noise_stack, ddim=False, guided=False, cond_grad_weight=1.0, trace=False, prepared=None, denom_items=None,
            defer_status=False, lowprec=None):
        """p_sample_loop / ddim_sample_loop (gaussian_diffusion.py:391-508 / :618-718) in one native call.
        Returns the reference's dict(sample, pred_xstart, other_outputs)."""
        m = self.model
        with _lib.on_device(m.device):
            try:
                return self._run_on_device(diffusion, batch, noise_stack, ddim, guided, cond_grad_weight, trace, prepared, denom_items, defer_status, lowprec)
            except _lib.EgoHMRRangeError:
                # an activation left the f16 range and was clamped (status bit 2 of the handle, raised by the conv kernels' stores): never silent.
                # on_saturation = 'f32': this checkpoint gets float32 activations from now on (exact-f32 MFMA path, ~3x slower) and the call runs again
                if m.on_saturation != "f32" or m.gcn_precision == "f32":
                    raise
                import warnings
                warnings.warn("egohmr_amd: a denoiser activation reached the f16 range (|x| >= 65504) and was clamped in the split-f16 / f16 path; "
                              "EgoHMR.on_saturation = 'f32': switching this model to gcn_precision = 'f32' and re-running the call", RuntimeWarning)
                m.gcn_precision = "f32"
                return self._run_on_device(diffusion, batch, noise_stack, ddim, guided, cond_grad_weight, trace, prepared, denom_items, False, None)

    def _run_on_device(self, diffusion, batch, noise_stack, ddim, guided, cond_grad_weight, trace, prepared, denom_items, defer_status, lowprec):
        m, L = self.model, _lib.lib()
        nonlocal_ci = m.diffusion_model.non_local.inter_channels if m.diffusion_model.nonlocal_layer else 0
        if nonlocal_ci and m.gcn_precision == "f16":
            raise _lib.EgoHMRHipError("the optional non-local GCN block runs on float32 features; use gcn_precision 'f16x3' or 'f32' with it")
        ev = getattr(self, "_status_event", None)
        if ev is not None and ev.query():                 # a deferred status word of an earlier call has arrived: look at it now
            self.check_status()
        st = prepared if prepared is not None else self.prepare(batch)
        B, T, hid, V = st.B, diffusion.num_timesteps, m.diffusion_model.hid_dim, m.smpl.num_verts
        noise = _lib.f32(noise_stack, m.device)
        assert noise.shape[0] >= T + 1 and noise.shape[1] == B and noise.shape[2] == 144, noise.shape
        steps, first_guided = self.step_table(diffusion, ddim, cond_grad_weight, guided)
        any_guided = first_guided < T
        n_guided = T - first_guided                       # (the reference guides a contiguous tail: t < 10, gaussian_diffusion.py:378-385)
        tvecs = self.timestep_vectors([diffusion.timestep_map[i] for i in range(T - 1, -1, -1)])   # [T,2,hid]
        passes = 2 if m.diffuse_fuse else 1
        # precision schedule: an explicit `lowprec` (calibration runs), else EgoHMR.f16x3_last_steps; 'auto' = the k calibrated for THESE
        # weights and THIS sampler - measured now, on this batch's first items, when it is not cached yet (auto_calibrate) - or no f16 step
        if nonlocal_ci:
            lowprec = 0                                   # the block reads float32 features: no plain-f16 steps
        if lowprec is None:
            skey = None
            if m.f16x3_last_steps == "auto" and m.gcn_precision == "f16x3":
                skey = self.schedule_key(diffusion, ddim, n_guided, cond_grad_weight, self.guide_denom(denom_items or B))
                if skey not in self._sched_cache and m.auto_calibrate:
                    self.calibrate_schedule(diffusion, batch, ddim=ddim, guided=guided, cond_grad_weight=cond_grad_weight, prepared=st,
                                            denom_items=denom_items or B, n_guided=n_guided)
                self.schedule_info = self._sched_cache.get(skey)
            lowprec = self.lowprec_steps(T, n_guided, ddim, key=skey)
        self.last_lowprec = int(lowprec)                  # leading steps of THIS call on plain f16 operands
        _, num_masked = self._apply_pass_map(st, passes)
        desc = _lib.SampleDesc(B=B, passes=passes, num_steps=T, ddim=int(ddim), per_step_launches=int(bool(m.per_step_launches)),
                               lbs_every_step=int(m.lbs_every_step), num_scene_points=st.scene.shape[1] if any_guided else 0,
                               guide_denom=self.guide_denom(denom_items or B), tau=m.collision_tau, num_masked=num_masked,
                               guide_all_points=int(bool(m.guide_all_points)), lowprec_steps=int(lowprec), nonlocal_ci=int(nonlocal_ci))
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
                # (every pointer the captured launches bake in that is not inside `bufs`: the two native handles and the mean / std buffers)
                key = (B, T, int(ddim), desc.passes, desc.lbs_every_step, desc.lowprec_steps, m.gcn_precision, self._gcn_key,
                       bytes(steps), st.scene.shape[1], num_masked, smpl_h.value if hasattr(smpl_h, "value") else int(smpl_h or 0),
                       mean.data_ptr(), std.data_ptr(), self._folded.Wx.data_ptr(), getattr(self, '_nl_set', None))
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
        # non-finite noise: x_T or a step's draw poisons the item from the next denoiser evaluation on; the LAST step's draw is multiplied by
        # nonzero_mask = 0 (DDIM: by sigma = 0 too) and 0 * NaN = NaN lands in that element of `sample` only (gaussian_diffusion.py:357-359, :575-580)
        self.last_trace = tr
        if tr is not None:
            batch["x_t"] = tr[-1]
        batch["vis_mask_smpl"] = st.vis_bool
        out = m._pack_output(batch, st, x0, pose6d, R, verts, joints, chk=noise, chk_rows=T, last_noise=noise[T], x_final=x_final)
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
                m.backbone.check_status()                 # (the stream has been waited for: the trunk's stream-K time-out word is there)
        # ddim_sample_with_grad hands out the GUIDED x0 of its last step as pred_xstart (gaussian_diffusion.py:587-592) while other_outputs keep the model's own;
        # that step has alpha_bar_prev = 1, so its sample IS the guided x0 (x0g * 1 + 0 * eps)
        return {"sample": x_final, "pred_xstart": x_final if (ddim and any_guided) else x0, "other_outputs": out}

    def check_status(self):
        """Raise if a sampling call issued with defer_status=True flagged its chained launches (see run()), or if a stream-K conv of the ResNet-50
        trunk timed out in a hand-off (ResNet50Features.check_status).  Waits for that call."""
        self.model.backbone.check_status()
        ev = getattr(self, "_status_event", None)
        if ev is None:
            return
        ev.synchronize()
        self._status_event = None
        if int(self._status_host[0]) != 0:
            self._status_host.zero_()
            with torch.cuda.device(self.model.device):
                _lib.check(_lib.lib().ehm_gcn_stack_status(self.gcn(), _lib.stream_ptr()), "ehm_gcn_stack_status")
