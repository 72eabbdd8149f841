"""Step-invariant conditioning encoders (run once per item, SURVEY.md section 0 finding 2).

The reference re-evaluates both inside every denoising step (models/egohmr/egohmr.py:183,:214);
neither depends on x_t or t, so the build evaluates them once per sampled batch.  Both run on the
hand-written split-f16 matrix-core kernels (csrc/conv.hip: the 52 bottleneck convolutions of ResNet-50
as NHWC implicit GEMMs with BatchNorm folded; csrc/linear.hip: the scene PointNet's GEMMs with the
max-pool fused; csrc/stem.hip: the 7x7 stem + ReLU + max-pool as one vector-ALU kernel).  Sub-module and
parameter names follow the reference so its checkpoints load (``backbone.*`` models/resnet.py:97-136,
``scene_enc.*`` models/respointnet.py:13-27).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib


# measurement hook (tools/exp_encoder_precision.py): called with every X2 activation buffer an encoder kernel has just written, (buffer, channels).
# None on the product path.
_x2_debug_hook = None


class _Bottleneck(nn.Module):
    def __init__(self, cin, width, stride, project):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, width * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(width * 4)
        self.downsample = None
        if project:
            self.downsample = nn.Sequential(nn.Conv2d(cin, width * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(width * 4))

    def forward(self, x):
        raise _lib.EgoHMRHipError("_Bottleneck is a parameter container: the arithmetic runs in ResNet50Features.folded() (csrc/conv.hip); "
                                  "egohmr_amd has no eager / CPU route (tools/_eager.py holds the eager yardstick)")


class ResNet50Features(nn.Module):
    """models/resnet.py:139-150: ResNet-50 trunk, global average pool -> [B,2048]."""

    hi_only = False      # True: the plain-f16 tier of the trunk (ehm_conv_x2_desc.hi_only: hi halves only, one MFMA per product) - NOT parity grade

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for i, (width, n, stride) in enumerate([(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)], 1):
            blocks = []
            for b in range(n):
                blocks.append(_Bottleneck(cin, width, stride if b == 0 else 1, project=(b == 0)))
                cin = width * 4
            setattr(self, f"layer{i}", nn.Sequential(*blocks))

    def current(self):
        """folded() for the weights as they are now: rebuilt when a parameter / buffer changes (storage or version counter)."""
        if getattr(self, "_fold_key_fn", None) is None:
            self._fold_key_fn = _lib.TensorKey(self)
        key = self._fold_key_fn()
        if getattr(self, "_fold_key", None) != key:
            self._fold_fn, self._fold_key = self.folded(), key
        return self._fold_fn

    def forward(self, x):
        """The module call IS the HIP path (models/resnet.py:139-150 in eval mode); there is no eager route."""
        return self.current()(x)

    # ------------------------------------------------------------------ stream-K hand-off time-outs (csrc/conv.hip): made loud, never waited for
    def _sk_status_async(self, ws):
        """Behind a trunk pass: a stream-ordered copy of the workspace's time-out count to pinned memory + an event.  An earlier pass's word
        that has arrived is looked at first (no host wait on the product path)."""
        if ws is None:                                     # (no conv of this pass had a stream-K plan)
            return
        ev = getattr(self, "_sk_event", None)
        if ev is not None and ev.query():
            self.check_status()
        if getattr(self, "_sk_host", None) is None:
            self._sk_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        _lib.check(_lib.lib().ehm_conv_x2_workspace_status(ws.data_ptr(), self._sk_host.data_ptr(), _lib.stream_ptr()), "ehm_conv_x2_workspace_status")
        self._sk_event = torch.cuda.Event()
        self._sk_event.record()
        self._sk_ws_checked = ws

    def check_status(self):
        """Raise if a stream-K conv of an earlier trunk pass timed out waiting for a partner block's partial sums (its tiles are NaN and the
        workspace's counters poisoned); the workspace is zeroed again by the call, so the next pass is clean.  Waits for that pass."""
        ev = getattr(self, "_sk_event", None)
        if ev is None:
            return
        ev.synchronize()
        self._sk_event = None
        if int(self._sk_host[0]) != 0:
            self._sk_host.zero_()
            ws = self._sk_ws_checked
            with torch.cuda.device(ws.device):
                _lib.check(_lib.lib().ehm_conv_x2_workspace_status(ws.data_ptr(), None, _lib.stream_ptr()), "ehm_conv_x2_workspace_status")

    # ------------------------------------------------------------------ inference form: BatchNorm folded into the convolutions
    @torch.no_grad()
    def folded(self, x2_activations: bool = True, fuse_shortcut: bool = True):
        """Eval-mode ResNet-50 on the matrix cores with every BatchNorm2d folded into its convolution (w' = w * g/sqrt(v+eps),
        b' = beta - mean * g/sqrt(v+eps); exact up to float re-association).  x2_activations=True (the product path): the activations
        stay in the X2 split format from the stem to the average pool (ehm_conv_x2); False: float32 NHWC activations between the
        convolutions (ehm_conv_nhwc_split - the kernel the non-local GCN block also uses; kept as a second implementation the tests
        compare).  HIP tensors only: there is no eager / library-convolution route."""
        import ctypes as C
        import math

        from . import _lib

        def fold(conv, bn):
            scale = (bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps))
            w = (conv.weight.double() * scale.view(-1, 1, 1, 1)).float()
            b = (bn.bias.double() - bn.running_mean.double() * scale).float()
            return w, b, conv.stride, conv.padding

        stem = fold(self.conv1, self.bn1)
        blocks = []
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                blocks.append((fold(blk.conv1, blk.bn1), fold(blk.conv2, blk.bn2), fold(blk.conv3, blk.bn3),
                               fold(blk.downsample[0], blk.downsample[1]) if blk.downsample is not None else None))
        packed = {}
        sk_ws = {}                                         # device -> scratch of the stream-K convs (one conv at a time on a stream)

        def pack(p):
            """weights [Co,Ci,KH,KW] -> tap-major [Co_pad, KH*KW*Ci] X2 split format, scaled by a power of two"""
            w = p[0]
            if id(w) in packed:
                return packed[id(w)]
            Co, Ci, KH, KW = w.shape
            K = KH * KW * Ci
            Co_pad = (Co + 127) // 128 * 128
            w2 = torch.zeros(Co_pad, K, device=w.device)
            w2[:Co] = w.permute(0, 2, 3, 1).reshape(Co, K)
            amax = float(w2.abs().max())
            scale = 2.0 ** math.floor(math.log2(2048.0 / amax)) if amax > 0 else 1.0
            buf = torch.empty(Co_pad, K, device=w.device)              # X2 rows have the byte size of float rows
            _lib.check(_lib.lib().ehm_split_pack(w2.data_ptr(), buf.data_ptr(), Co_pad, K, K, scale, _lib.stream_ptr()), "ehm_split_pack")
            packed[id(w)] = (buf, scale, p[1].contiguous(), (Co, Ci, KH, KW), p[2][0], p[3][0])
            return packed[id(w)]

        def conv_mc(x, p, res=None, relu=True):
            buf, scale, bias, (Co, Ci, KH, KW), stride, pad = pack(p)
            N, H, W, _ = x.shape
            Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
            y = torch.empty(N, Ho, Wo, Co, device=x.device)
            d = _lib.ConvDesc(x.data_ptr(), buf.data_ptr(), bias.data_ptr(), res.data_ptr() if res is not None else None, y.data_ptr(),
                              N, H, W, Ci, Co, KH, KW, stride, pad, 1 if relu else 0, scale)
            _lib.check(_lib.lib().ehm_conv_nhwc_split(C.byref(d), _lib.stream_ptr()), "ehm_conv_nhwc_split")
            return y

        stem_wt = stem[0].reshape(64, 147).t().contiguous()                        # [147][64], k = (ci*7 + kh)*7 + kw
        stem_b = stem[1].contiguous()

        def x2_buffer(pixels, ch, dev, clear_last=False):
            """X2 activation matrix for ehm_conv_x2: pixels rounded up to the row tile + one all-zero row (out-of-image taps read it;
            ehm_conv_x2 clears it in its output, the stem's buffer is cleared here)"""
            rows = int(_lib.lib().ehm_conv_x2_rows(pixels))
            buf = torch.empty(rows, ch, device=dev)                                 # X2 rows have the byte size of float rows
            if clear_last:
                buf[rows - 1].zero_()
            return buf

        def stem_mc(x, x2=False):
            """conv1 + bn1 + relu + maxpool in one pass, NCHW in -> NHWC out (csrc/stem.hip); x2: output in the X2 split format"""
            N, _, H, W = x.shape
            lib = _lib.lib()
            if stem_wt.device != x.device:
                raise _lib.EgoHMRHipError("ResNet50Features.folded(): weights and input live on different devices")
            scratch = torch.empty(lib.ehm_resnet_stem_scratch_bytes(N, H, W) // 4, device=x.device)
            y = x2_buffer(N * (H // 4) * (W // 4), 64, x.device, clear_last=True) if x2 else torch.empty(N, H // 4, W // 4, 64, device=x.device)
            _lib.check(lib.ehm_resnet_stem(x.data_ptr(), stem_wt.data_ptr(), stem_b.data_ptr(), scratch.data_ptr(), y.data_ptr(), N, H, W,
                                           1 if x2 else 0, _lib.stream_ptr()), "ehm_resnet_stem")
            if x2 and _x2_debug_hook is not None:
                _x2_debug_hook(y, 64)
            return y

        def pack_dual(p, ds):
            """a bottleneck's last conv and its projection shortcut as ONE weight matrix [Co_pad, Ci + Ci_in] (both 1 x 1), common scale, summed bias"""
            key = (id(p[0]), id(ds[0]))
            if key in packed:
                return packed[key]
            w, wd = p[0], ds[0]
            Co, Ci = w.shape[:2]
            assert w.shape[2:] == (1, 1) and wd.shape[2:] == (1, 1) and wd.shape[0] == Co and Co % 128 == 0
            K = Ci + wd.shape[1]
            w2 = torch.cat([w.reshape(Co, Ci), wd.reshape(Co, -1)], dim=1).contiguous()
            amax = float(w2.abs().max())
            scale = 2.0 ** math.floor(math.log2(2048.0 / amax)) if amax > 0 else 1.0
            buf = torch.empty(Co, K, device=w.device)
            _lib.check(_lib.lib().ehm_split_pack(w2.data_ptr(), buf.data_ptr(), Co, K, K, scale, _lib.stream_ptr()), "ehm_split_pack")
            packed[key] = (buf, scale, (p[1].double() + ds[1].double()).float().contiguous(), (Co, Ci, 1, 1), p[2][0], p[3][0], wd.shape[1], ds[2][0])
            return packed[key]

        def conv_x2(x, shape, p, res=None, relu=True, shortcut=None):
            """one bottleneck conv on X2 activations (csrc/conv.hip conv_x2_tile_kernel); shape = (N, H, W) of x.
            shortcut = (block input, its shape, folded downsample): the projection shortcut accumulates inside this conv (second K segment)."""
            N, H, W = shape
            if shortcut is not None:
                buf, scale, bias, (Co, Ci, KH, KW), stride, pad, Ci2, stride2 = pack_dual(p, shortcut[2])
            else:
                buf, scale, bias, (Co, Ci, KH, KW), stride, pad = pack(p)
            Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
            y = x2_buffer(N * Ho * Wo, Co, x.device)
            d = _lib.ConvX2Desc(x.data_ptr(), x.shape[0], buf.data_ptr(), bias.data_ptr(), res.data_ptr() if res is not None else None, y.data_ptr(),
                                N, H, W, Ci, Co, KH, KW, stride, pad, 1 if relu else 0, scale, None, 0)
            d.hi_only = int(bool(self.hi_only))                                  # the plain-f16 tier (EgoHMR.encoder_precision = 'f16'): NOT parity grade
            if shortcut is not None:
                x_in, (_, H2, W2) = shortcut[0], shortcut[1]
                d.x2, d.x2_rows, d.H2, d.W2, d.Ci2, d.stride2 = x_in.data_ptr(), x_in.shape[0], H2, W2, Ci2, stride2
            need = int(_lib.lib().ehm_conv_x2_workspace_bytes(C.byref(d)))      # stream-K scratch (layers 3 / 4: tile counts that straddle the block slots)
            if need:
                ws = sk_ws.get(str(x.device))
                if ws is None or ws.numel() < need:
                    ws = sk_ws[str(x.device)] = torch.zeros(need, dtype=torch.uint8, device=x.device)   # zeroed ONCE: the convs leave their counters zeroed
                d.workspace, d.workspace_bytes, d.workspace_clean = ws.data_ptr(), ws.numel(), 1
            _lib.check(_lib.lib().ehm_conv_x2(C.byref(d), _lib.stream_ptr()), "ehm_conv_x2")
            if _x2_debug_hook is not None:
                _x2_debug_hook(y, Co)
            return y, (N, Ho, Wo)

        def run_x2(x):
            """the whole trunk with the activations in the X2 split format between the layers (taps gathered by the LDS DMA)"""
            N = x.shape[0]
            shp = (N, x.shape[2] // 4, x.shape[3] // 4)
            x = stem_mc(x, x2=True)
            for c1, c2, c3, ds in blocks:
                y, s1 = conv_x2(x, shp, c1)
                y, s2 = conv_x2(y, s1, c2)
                if ds is None:
                    x, shp = conv_x2(y, s2, c3, res=x)
                elif fuse_shortcut:
                    x, shp = conv_x2(y, s2, c3, shortcut=(x, shp, ds))          # out = conv3(.) + downsample(x) in one launch: no shortcut tensor
                else:
                    x, shp = conv_x2(y, s2, c3, res=conv_x2(x, shp, ds, relu=False)[0])
            out = torch.empty(N, x.shape[1], device=x.device)
            _lib.check(_lib.lib().ehm_x2_group_mean(x.data_ptr(), out.data_ptr(), N, shp[1] * shp[2], x.shape[1], int(bool(self.hi_only)), _lib.stream_ptr()),
                       "ehm_x2_group_mean")
            self._sk_status_async(sk_ws.get(str(x.device)))
            return out

        def run(x):
            if not x.is_cuda:
                raise _lib.EgoHMRHipError("the ResNet-50 backbone needs its input on a HIP device; egohmr_amd has no CPU path")
            x = _lib.f32(x)
            if x.shape[1] != 3 or x.shape[2] % 32 or x.shape[3] % 32:
                raise _lib.EgoHMRHipError(f"the fused stem needs [N,3,H,W] images with H, W multiples of 32 (the reference feeds 224 x 224 crops); got {tuple(x.shape)}")
            with _lib.on_device(x.device):                                          # the library launches on the CURRENT device's stream
                if x2_activations:
                    return run_x2(x)
                x = stem_mc(x)                                                      # NHWC float32 from here on
                for c1, c2, c3, ds in blocks:
                    y = conv_mc(conv_mc(x, c1), c2)
                    x = conv_mc(y, c3, res=x if ds is None else conv_mc(x, ds, relu=False))
                return x.mean(dim=(1, 2))

        return run


class _ResBlockFC(nn.Module):
    def __init__(self, cin, cout, hidden):
        super().__init__()
        self.fc_0 = nn.Linear(cin, hidden)
        self.fc_1 = nn.Linear(hidden, cout)
        self.shortcut = nn.Linear(cin, cout, bias=False)

    def forward(self, x):
        raise _lib.EgoHMRHipError("_ResBlockFC is a parameter container: ResnetPointnet.forward runs the block on the HIP kernels (csrc/linear.hip)")


class ResnetPointnet(nn.Module):
    """models/respointnet.py:33-59: per-point MLP with three global max-pool-concat stages -> [B,out_dim].

    Parameters keep the reference's names; the arithmetic runs on the split-f16 matrix-core kernels of
    csrc/linear.hip (f32-grade, see docs/EXPERIMENTS.md 3.3) with the per-body constant half of every block input folded
    into bias vectors, fc_1 + shortcut fused into one dual-source GEMM and the max-pool fused into its epilogue."""

    hi_only = False      # True: the plain-f16 tier (ehm_linear_desc.hi_only) - NOT parity grade

    def __init__(self, out_dim=512, hidden_dim=256):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.fc_pos_0 = nn.Linear(3, 2 * hidden_dim)
        for b in range(4):
            setattr(self, f"block_{b}", _ResBlockFC(2 * hidden_dim, hidden_dim, hidden_dim))
        self.fc_c = nn.Linear(hidden_dim, out_dim)
        self._packed = None
        self._packed_key = None

    # ------------------------------------------------------------------ weight preparation (once per weight version)
    @staticmethod
    def _pack(w64: torch.Tensor, device):
        """float64 [N,K] -> (X2 buffer, power-of-two scale); K padded to a multiple of 32."""
        from . import _lib
        w = w64.float().contiguous().to(device)
        N, K = w.shape
        Kp = (K + 31) // 32 * 32
        amax = float(w.abs().max())
        scale = 2.0 ** (12 - int(np.floor(np.log2(amax)) + 1)) if amax > 0 else 1.0
        buf = torch.empty(N, Kp, dtype=torch.float32, device=device)         # X2 has the byte size of float32 [N,Kp]
        _lib.check(_lib.lib().ehm_split_pack(w.data_ptr(), buf.data_ptr(), N, K, Kp, scale, _lib.stream_ptr()), "ehm_split_pack")
        return buf, scale, w

    def _prepare(self, device):
        if getattr(self, "_tkey", None) is None:
            from . import _lib
            self._tkey = _lib.TensorKey(self)
        key = self._tkey() + (str(device),)
        if self._packed is not None and self._packed_key == key:
            return self._packed
        H = self.hidden_dim
        d = lambda t: t.detach().double()
        P = {"pos_w4": torch.cat([self.fc_pos_0.weight.detach().float(), self.fc_pos_0.bias.detach().float()[:, None]], 1).contiguous()}   # rows (w_x, w_y, w_z, bias)
        blocks = [self.block_0, self.block_1, self.block_2, self.block_3]
        b0 = blocks[0]
        P["g1_0"] = self._pack(d(b0.fc_0.weight), device) + (b0.fc_0.bias.detach().float().contiguous(),)
        s_pos = d(b0.shortcut.weight) @ d(self.fc_pos_0.weight)                          # shortcut(fc_pos(p)) folded: [H,3]
        w = torch.cat([d(b0.fc_1.weight), torch.cat([s_pos, s_pos.new_zeros(H, 29)], 1)], dim=1)   # [H, H+32]
        P["g3_0"] = self._pack(w, device) + ((d(b0.fc_1.bias) + d(b0.shortcut.weight) @ d(self.fc_pos_0.bias)).float().contiguous(),)
        for i in (1, 2, 3):
            bl = blocks[i]
            W0, S = d(bl.fc_0.weight), d(bl.shortcut.weight)
            P[f"g1_{i}"] = self._pack(W0[:, :H], device)
            P[f"g3_{i}"] = self._pack(torch.cat([d(bl.fc_1.weight), S[:, :H]], dim=1), device) + (bl.fc_1.bias.detach().float().contiguous(),)
            # pooled halves: per-body bias vectors [relu(pooled) . W0b^T + b0 | pooled . Sb^T], [B,H] x [H,2H] in exact float32 and ONE launch
            # (ehm_skinny_gemm_f32 wants W as [K,N]; its `relu` argument rectifies the input for the first H output columns)
            P[f"wvsT_{i}"] = torch.cat([W0[:, H:].t(), S[:, H:].t()], dim=1).float().contiguous().to(device)
            b0v = bl.fc_0.bias.detach().float()
            P[f"bvs_{i}"] = torch.cat([b0v, torch.zeros_like(b0v)]).contiguous().to(device)
        P["fc_cT"], P["fc_cb"] = d(self.fc_c.weight).t().float().contiguous().to(device), self.fc_c.bias.detach().float().contiguous().to(device)
        self._packed, self._packed_key = P, key
        return P

    @torch.no_grad()
    def forward(self, p):
        from . import _lib
        if not p.is_cuda:
            raise _lib.EgoHMRHipError("ResnetPointnet runs on the HIP kernels only (got a CPU tensor); there is no CPU path")
        with _lib.on_device(p.device):                                                   # the library launches on the CURRENT device's stream
            return self._forward_on_device(p)

    def _forward_on_device(self, p):
        from . import _lib
        L, dev, H = _lib.lib(), p.device, self.hidden_dim
        P = self._prepare(dev)
        p = _lib.f32(p)
        B, N, _ = p.shape
        Np = (N + 191) // 192 * 192                                                      # row tile of csrc/linear.hip
        M = B * Np
        st = _lib.stream_ptr()
        f32buf = lambda cols: torch.empty(M, cols, dtype=torch.float32, device=dev)      # X2 buffers (same bytes as float32)
        P32, Hb, netA, netB = f32buf(32), f32buf(H), f32buf(H), f32buf(H)
        p = p.contiguous()
        _lib.check(L.ehm_pointnet_lift(p.data_ptr(), None, None, None, P32.data_ptr(), B, N, Np, 2 * H, st), "ehm_pointnet_lift")

        def gemm(A0, K0, A1, K1, W, bias, gbias, Y, colmax, relu_in0, relu_out, lift=False, gstride=0):
            d = _lib.LinearDesc(A0=A0.data_ptr() if A0 is not None else None, A1=A1.data_ptr() if A1 is not None else None, W=W[0].data_ptr(),
                                lift_points=p.data_ptr() if lift else None, lift_W4=P["pos_w4"].data_ptr() if lift else None,
                                bias=bias.data_ptr() if bias is not None else None,
                                group_bias=gbias.data_ptr() if gbias is not None else None,
                                Y=Y.data_ptr() if Y is not None else None, colmax=colmax.data_ptr() if colmax is not None else None,
                                M=M, N=H, K0=K0, K1=K1, rows_per_group=Np, valid_rows_per_group=N, relu_in0=int(relu_in0),
                                relu_out=int(relu_out), w_scale=W[1], hi_only=int(bool(self.hi_only)), group_bias_stride=gstride)
            _lib.check(L.ehm_linear_split(d, st), "ehm_linear_split")
            if Y is not None and _x2_debug_hook is not None:
                _x2_debug_hook(Y, H)

        def small(x, Wt, bias, relu_in_cols=0):
            """[B,K] x [K,N] (+ bias) in exact float32: the per-body vectors between the big GEMMs.  The BLAS ran each of these
            256 x 256 x 256 products as one 256 x 256 workgroup (170 - 450 us, on the PointNet's critical path).
            relu_in_cols: the first that many output columns see relu(x)."""
            y = torch.empty(x.shape[0], Wt.shape[1], device=dev)
            _lib.check(L.ehm_skinny_gemm_f32(x.data_ptr(), Wt.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(),
                                             x.shape[0], Wt.shape[0], Wt.shape[1], relu_in_cols << 1, st), "ehm_skinny_gemm_f32")
            return y

        neg_inf = float("-inf")
        # block_0 on net0 = fc_pos(p):  h = fc_0(relu(net0));  net1 = fc_1(relu(h)) + shortcut(net0)
        # (relu(net0) is produced inside the GEMM's loader from the 12 bytes of each point: ehm_linear_desc.lift_points)
        gemm(None, 2 * H, None, 0, P["g1_0"], P["g1_0"][3], None, Hb, None, False, True, lift=True)
        pooled_all = torch.full((4, B, H), neg_inf, device=dev)                         # the four column maxima, cleared by one launch
        pooled = pooled_all[0]
        gemm(Hb, H, P32, 32, P["g3_0"], P["g3_0"][3], None, netA, pooled, False, False)
        cur, nxt = netA, netB
        for i in (1, 2, 3):
            # pooled halves of fc_0(relu(cat[net, pooled])) and of shortcut(cat[net, pooled]): one launch, [B, 2H]
            vs = small(pooled, P[f"wvsT_{i}"], P[f"bvs_{i}"], relu_in_cols=H)
            v, s = vs[:, :H], vs[:, H:]
            gemm(cur, H, None, 0, P[f"g1_{i}"], None, v, Hb, None, True, True, gstride=2 * H)
            pooled = pooled_all[i]
            gemm(Hb, H, cur, H, P[f"g3_{i}"], P[f"g3_{i}"][3], s, nxt if i < 3 else None, pooled, False, False, gstride=2 * H)
            cur, nxt = nxt, cur
        return small(pooled, P["fc_cT"], P["fc_cb"], relu_in_cols=P["fc_cT"].shape[1])

