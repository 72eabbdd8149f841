#!/usr/bin/env python3
"""What would the reference's OWN way of running this path - eager PyTorch, float32 - deliver on an MI355X?  The reference cannot be imported on the GPU
box (and needs smplx / coap); this script runs the same graph as plain torch ops on the GPU (MIOpen convolutions, hipBLASLt GEMMs) from the package's
state_dict: ResNet-50 + ResnetPointnet, the 2694-wide condition, the ModulatedGCN with two passes (diffuse_fuse), the DDPM posterior update, for
B = 256, 100 steps.  Two modes like bench.py's cpu_baseline: 'faithful' (egohmr.py:182-223 re-encodes image and scene inside EVERY model call) and
'hoisted' (encoders once).  The SMPL forward of every step is LEFT OUT of the eager timing (smplx is not here) - that favours the baseline.
Checks one denoiser evaluation against the package (pred_x_start) before timing.

    MIOPEN_FIND_MODE=FAST python tools/eager_gpu_yardstick.py [steps_timed]
"""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egohmr_amd import synthetic as syn  # noqa: E402
from egohmr_amd.diffusion import create_gaussian_diffusion  # noqa: E402
from egohmr_amd.factory import batch_to_device, build_synthetic_model  # noqa: E402
from _eager import resnet50_eager  # noqa: E402

dev = torch.device("cuda:0")
steps_timed = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, N, T = 256, 4096, 100
model = build_synthetic_model(dev, 0, sensitive=dict(num_diffusion_timesteps=T))
sd = {k: v.detach() for k, v in model.state_dict().items()}
adj = model.diffusion_model.adj.to(dev)
eye = torch.eye(24, device=dev)


def bn(x, p, eps=1e-5):                                      # eval-mode BatchNorm over dim 1
    shape = [1, -1] + [1] * (x.dim() - 2)
    return (x - sd[p + ".running_mean"].view(shape)) / torch.sqrt(sd[p + ".running_var"].view(shape) + eps) * sd[p + ".weight"].view(shape) + sd[p + ".bias"].view(shape)


def mgconv(p, x):                                            # modulated_gcn_conv.py:39-50
    W, M = sd[p + ".W"], sd[p + ".M"]
    a = adj + sd[p + ".adj2"]
    a = (a.T + a) / 2
    return torch.matmul(a * eye, M * torch.matmul(x, W[0])) + torch.matmul(a * (1 - eye), M * torch.matmul(x, W[1])) + sd[p + ".bias"].view(1, 1, -1)


def gconv(p, x):                                             # modulated_gcn.py:21-28
    return F.relu(bn(mgconv(p + ".gconv", x).transpose(1, 2), p + ".bn").transpose(1, 2))


def gcn(x, p="diffusion_model."):                            # modulated_gcn.py:99-116
    out = gconv(p + "gconv_input.0", x)
    for b in range(4):
        out = out + gconv(f"{p}gconv_layers.{b}.gconv2", gconv(f"{p}gconv_layers.{b}.gconv1", out))
    return mgconv(p + "gconv_output", out)


def pointnet(pts, p="scene_enc."):                           # respointnet.py:33-59
    lin = lambda n, v, bias=True: F.linear(v, sd[p + n + ".weight"], sd[p + n + ".bias"] if bias else None)

    def block(b, x):
        return lin(b + ".shortcut", x, False) + lin(b + ".fc_1", F.relu(lin(b + ".fc_0", F.relu(x))))
    x = block("block_0", lin("fc_pos_0", pts))
    for b in (1, 2, 3):
        x = block(f"block_{b}", torch.cat([x, x.max(dim=1, keepdim=True)[0].expand(x.size())], dim=2))
    return lin("fc_c", F.relu(x.max(dim=1)[0]))


def encode(batch):                                           # egohmr.py:182-223
    img_feats = resnet50_eager(model.backbone, batch["img"])   # plain eager torch (MIOpen): tools/_eager.py
    transl = batch["smpl_params"]["transl"]
    scene_feats = pointnet(batch["scene_pcd_verts_full"] - transl.unsqueeze(1))
    tf = F.linear(transl, sd["transl_enc.0.weight"], sd["transl_enc.0.bias"]) if "transl_enc.0.weight" in sd else model.transl_enc(transl)
    fx = batch["fx"]
    ofx = fx * model.cfg.CAM.FX_NORM_COEFF
    cam = torch.cat([torch.stack([batch["cam_cx"] / ofx, batch["cam_cy"] / ofx], -1),
                     torch.stack([batch["box_center"][:, 0] / ofx, batch["box_center"][:, 1] / ofx, batch["box_size"] / ofx], -1), fx.unsqueeze(1)], dim=1)
    return img_feats, torch.cat([scene_feats, tf, cam], dim=1)


def denoise(batch, x_t, t_model, enc):                       # egohmr.py:173-258 (x0 only)
    img_feats, other = enc
    vis = model.visibility(batch)
    e = sd["embed_timestep.sequence_pos_encoder.pe"][t_model][:, 0]
    e = F.linear(F.silu(F.linear(e, sd["embed_timestep.time_embed.0.weight"], sd["embed_timestep.time_embed.0.bias"])),
                 sd["embed_timestep.time_embed.2.weight"], sd["embed_timestep.time_embed.2.bias"])
    temb = e.unsqueeze(1).repeat(1, 24, 1)
    img24 = img_feats.unsqueeze(1).repeat(1, 24, 1) * vis.unsqueeze(-1).float()
    cond = torch.cat([img24, other.unsqueeze(1).repeat(1, 24, 1)], dim=-1)
    x_feat = F.linear(x_t.reshape(B, 24, -1), sd["input_process.poseEmbedding.weight"], sd["input_process.poseEmbedding.bias"])
    out_c = gcn(torch.cat([cond, x_feat, temb], dim=-1))
    cond_u = cond.clone()
    cond_u[:, :, 0:2048] = 0
    out = gcn(torch.cat([cond_u, x_feat, temb], dim=-1)).reshape(B, -1)
    m = vis.unsqueeze(-1).repeat(1, 1, 6).reshape(B, -1)
    out[m] = out_c.reshape(B, -1)[m]
    return out


with torch.no_grad():
    batch = batch_to_device(syn.make_batch(B, N, seed=0), dev)
    d = create_gaussian_diffusion(num_diffusion_timesteps=T, timestep_respacing="")
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(B, 144, device=dev, generator=g)
    enc = encode(batch)
    t_chk = torch.full((B,), 37, device=dev, dtype=torch.long)
    x0_eager = denoise(batch, x, t_chk, enc)
    batch["x_t"] = x
    x0_pkg = model(batch, t_chk)["pred_x_start"]
    err = float((x0_eager - x0_pkg).abs().max() / x0_pkg.abs().max())
    assert err < 1e-3, err
    c1 = torch.as_tensor(d.posterior_mean_coef1, device=dev, dtype=torch.float32)
    c2 = torch.as_tensor(d.posterior_mean_coef2, device=dev, dtype=torch.float32)
    lv = torch.as_tensor(d.posterior_log_variance_clipped, device=dev, dtype=torch.float32)

    def loop(n_steps, faithful):
        xt = x.clone()
        e = enc
        for i in range(T - 1, T - 1 - n_steps, -1):
            t = torch.full((B,), i, device=dev, dtype=torch.long)
            if faithful:
                e = encode(batch)
            x0 = denoise(batch, xt, t, e)
            xt = c1[i] * x0 + c2[i] * xt + (0.0 if i == 0 else 1.0) * torch.exp(0.5 * lv[i]) * torch.randn_like(xt)
        return xt

    for faithful in (False, True):
        n = steps_timed if not faithful else max(2, steps_timed // 5)
        loop(2, faithful)
        torch.cuda.synchronize()
        t0 = time.time()
        loop(n, faithful)
        torch.cuda.synchronize()
        per_step = (time.time() - t0) / n
        t0 = time.time()
        encode(batch)
        torch.cuda.synchronize()
        t_enc = time.time() - t0
        call = per_step * T + (0.0 if faithful else t_enc)
        print(json.dumps({"what": "eager PyTorch float32 on the GPU (MIOpen / hipBLASLt), B256 DDPM-100, 2 GCN passes, NO SMPL forward in the steps",
                          "mode": "faithful (encoders inside every step)" if faithful else "hoisted (encoders once)", "steps_timed": n, "ms_per_step": round(per_step * 1e3, 2),
                          "encoders_ms": round(t_enc * 1e3, 1), "ms_per_call_extrapolated": round(call * 1e3, 1), "bodies_per_s": round(B / call, 1),
                          "denoiser_check_rel_err_vs_package": err}), flush=True)
