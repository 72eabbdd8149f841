#!/usr/bin/env python3
"""Per-(step, hidden layer) precision freedom of the denoiser: how far do the final bodies move when ONE hidden conv - or a set of
them - multiplies on the hi halves only (plain-f16 operands: one MFMA per product instead of three) while everything else stays split-f16?

No new kernel is needed to ask the question: a conv on operands whose lo halves are ZERO computes exactly the hi-only products with the shipped
three-MFMA kernel (the two cross terms vanish).  So the sweep drives the sampling loop itself - per-conv launches through the C ABI, the product's own
input / output / sampler-step kernels - and, for the convs under test, feeds `ehm_gcn_hidden_layer` a copy of the activations with the lo halves
cleared and a second handle whose hidden weights were rounded to f16 (lo = 0 after the split).  Activation STORAGE stays X2 (hi + lo) everywhere: this
is the arithmetic a per-layer hi-only mode of the chain kernel would have.

    python tools/layer_precision_sweep.py [--batch 32] [--T 100] [--respacing ''] [--seed 0] [--tol 1e-4] [--out file.jsonl]

Rows (JSON lines): the loop's agreement with FusedSampler.run (sanity), every single layer hi-only in all steps, all layers hi-only in all steps, every
single layer hi-only in the first T - k steps, and a greedy set: layers added in order of increasing damage (all steps) while the max vertex distance to
the all-split loop stays below --tol; with the share of the hidden convs' MFMA work the set would save.
"""
import argparse
import copy
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--T", type=int, default=100)
    ap.add_argument("--respacing", default="")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--tol", type=float, default=1e-4)
    ap.add_argument("--weights", default="sensitive", choices=["sensitive", "insensitive"])
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    from egohmr_amd import _lib
    from egohmr_amd import synthetic as syn
    from egohmr_amd.diffusion import create_gaussian_diffusion
    from egohmr_amd.factory import batch_to_device, build_synthetic_model
    from egohmr_amd.fused import PRECISIONS
    dev = torch.device("cuda:0")
    L = _lib.lib()
    n_orig = a.T
    model = build_synthetic_model(dev, a.seed, diffuse_fuse=True, sensitive=dict(num_diffusion_timesteps=n_orig) if a.weights == "sensitive" else None)
    model.f16x3_last_steps = None
    model.gcn_precision = "f16x3"
    d = create_gaussian_diffusion(num_diffusion_timesteps=n_orig, timestep_respacing=a.respacing)
    T, B, ddim = d.num_timesteps, a.batch, bool(a.respacing)
    batch = batch_to_device(syn.make_batch(B, 1024, seed=100 + a.seed), dev)
    noise = torch.from_numpy(syn.make_noise_stack(T, B, seed=100 + a.seed)).to(dev)
    fs = model.fused_sampler
    ref_run = fs.run(d, dict(batch), noise, ddim=ddim)["other_outputs"]["pred_vertices"].clone()
    st = fs.prepare(batch)
    passes = 2 if model.diffuse_fuse else 1
    hA = fs.gcn()
    vb, _ = fs._apply_pass_map(st, passes)
    dm = model.diffusion_model
    hid, nl = dm.hid_dim, 2 * dm.num_layers
    # handle B: the same denoiser with hidden weights rounded to f16 -> their split has lo = 0
    dmB = copy.deepcopy(dm)
    with torch.no_grad():
        for blk in dmB.gconv_layers:
            for gc in (blk.gconv1.gconv, blk.gconv2.gconv):
                gc.W.copy_(gc.W.half().float())
    hB, keepB = dmB.create_native_handle(dev)
    _lib.check(L.ehm_gcn_set_precision(hB, PRECISIONS["f16x3"]))
    tile = L.ehm_gcn_row_tile()
    rows = vb * 24
    rows_pad = (rows + tile - 1) // tile * tile
    steps, _ = fs.step_table(d, ddim, 1.0, False)
    tvecs = fs.timestep_vectors([d.timestep_map[i] for i in range(T - 1, -1, -1)])
    mean, std = model._std_mean()
    s = None
    X = [torch.zeros(rows_pad, hid, device=dev) for _ in range(3)]
    Xc = torch.zeros(rows_pad, hid, device=dev)

    def lo_cleared(src):
        Xc.copy_(src)
        Xc.view(torch.int16).view(rows_pad, hid // 32, 2, 32)[:, :, 1, :] = 0          # X2<32>: 32 hi halves | 32 lo halves per group of 32 channels
        return Xc

    def loop(flags):
        """flags(k, l) -> bool: executed step k (0 = first = noisiest), hidden conv l multiplies hi-only."""
        x = noise[0].clone()
        x0 = torch.empty(B, 144, device=dev)
        xn = torch.empty(B, 144, device=dev)
        for k in range(T):
            c = steps[k]
            _lib.check(L.ehm_gcn_input_layer(hA, _lib.ptr(st.h_img), _lib.ptr(st.h_oth), _lib.ptr(st.vis), _lib.ptr(x), _lib.ptr(fs._folded.Wx),
                                             _lib.ptr(tvecs[k]), _lib.ptr(X[0]), B, passes, s), "input")
            cur = 0
            for blk in range(nl // 2):
                y2 = 2 if cur == 0 else 0
                for l, src, res, dst in ((2 * blk, X[cur], None, X[1]), (2 * blk + 1, X[1], X[cur], X[y2])):
                    if flags(k, l):
                        _lib.check(L.ehm_gcn_hidden_layer(hB, l, lo_cleared(src).data_ptr(), res.data_ptr() if res is not None else None, dst.data_ptr(), rows_pad, s), "hidden B")
                    else:
                        _lib.check(L.ehm_gcn_hidden_layer(hA, l, src.data_ptr(), res.data_ptr() if res is not None else None, dst.data_ptr(), rows_pad, s), "hidden A")
                cur = y2
            _lib.check(L.ehm_gcn_output_layer(hA, X[cur].data_ptr(), _lib.ptr(st.vis), x0.data_ptr(), B, passes, s), "output")
            nz = noise[1 + k]
            if ddim:
                _lib.check(L.ehm_ddim_step(x.data_ptr(), x0.data_ptr(), nz.data_ptr(), xn.data_ptr(), c.sqrt_recip_ac, c.sqrt_recipm1_ac, c.sqrt_ac_prev, c.dir_coef,
                                           c.sigma, c.nonzero, B * 144, s), "ddim")
            else:
                _lib.check(L.ehm_ddpm_step(x.data_ptr(), x0.data_ptr(), nz.data_ptr(), None, xn.data_ptr(), c.coef1, c.coef2, c.log_variance, c.nonzero, 0.0,
                                           B * 144, s), "ddpm")
            x, xn = xn, x
        verts = torch.empty(B, model.smpl.num_verts, 3, device=dev)
        joints = torch.empty(B, model.smpl.num_joints_out, 3, device=dev)
        _lib.check(L.ehm_smpl_forward_rot6d(model.smpl.handle(), _lib.ptr(st.betas), x0.data_ptr(), _lib.ptr(mean), _lib.ptr(std), verts.data_ptr(), joints.data_ptr(),
                                            None, None, None, B, s), "smpl")
        L.ehm_gcn_stack_status(hA, s)
        L.ehm_gcn_stack_status(hB, s)
        return verts

    out = open(a.out, "w") if a.out else None

    def emit(row):
        row = dict(T=T, B=B, respacing=a.respacing, weights=a.weights, seed=a.seed, **row)
        print(json.dumps(row), flush=True)
        if out:
            out.write(json.dumps(row) + "\n")

    base = loop(lambda k, l: False)
    dist = lambda v: float((v - base).norm(dim=-1).max())
    emit({"what": "sanity: this loop (all convs split-f16) vs FusedSampler.run", "max_vertex_dist_m": float((base - ref_run).norm(dim=-1).max())})
    single = {}
    for l in range(nl):
        single[l] = dist(loop(lambda k, ll, l=l: ll == l))
        emit({"what": "one layer hi-only, all steps", "layer": l, "max_vertex_dist_m": single[l]})
    emit({"what": "all layers hi-only, all steps (= f16 operands, X2 storage)", "max_vertex_dist_m": dist(loop(lambda k, l: True))})
    for keep in sorted({min(T, 10), min(T, 30), T // 2}):
        for l in range(nl):
            emit({"what": "one layer hi-only in the first T - k steps", "layer": l, "k_last_steps_split": keep,
                  "max_vertex_dist_m": dist(loop(lambda k, ll, l=l: ll == l and k < T - keep))})
        emit({"what": "all layers hi-only in the first T - k steps (= the per-step schedule with X2 storage)", "k_last_steps_split": keep,
              "max_vertex_dist_m": dist(loop(lambda k, l: k < T - keep))})
    # greedy set over whole layers (all steps)
    order = sorted(range(nl), key=lambda l: single[l])
    chosen, err = [], 0.0
    for l in order:
        trial = chosen + [l]
        e = dist(loop(lambda k, ll: ll in trial))
        emit({"what": "greedy trial (all steps)", "layers": trial, "max_vertex_dist_m": e, "accepted": e < a.tol})
        if e < a.tol:
            chosen, err = trial, e
        else:
            break
    # MFMA work of the hidden convs saved: a hi-only conv issues 1/3 of the matrix instructions
    emit({"what": "greedy result", "tol_m": a.tol, "layers_hi_only_all_steps": chosen, "max_vertex_dist_m": err,
          "hidden_conv_mfma_work_saved": len(chosen) / nl * (2.0 / 3.0)})
    # per-(step, layer): with the greedy set fixed, how many EARLY steps can run everything hi-only on top of it?
    best_k = T
    for keep in (T // 2, T // 4, T // 10, 0):
        e = dist(loop(lambda k, l, keep=keep: l in chosen or k < T - keep))
        emit({"what": "greedy set + all layers hi-only in the first T - k steps", "k_last_steps_split": keep, "max_vertex_dist_m": e, "within_tol": e < a.tol})
        if e < a.tol:
            best_k = keep
        else:
            break
    frac = (len(chosen) / nl + (1 - len(chosen) / nl) * (T - best_k) / T) * (2.0 / 3.0)
    emit({"what": "combined (step, layer) schedule", "tol_m": a.tol, "layers_hi_only_all_steps": chosen, "other_layers_hi_only_first_steps": T - best_k,
          "hidden_conv_mfma_work_saved": frac})
    L.ehm_gcn_destroy(hB)
    if out:
        out.close()


if __name__ == "__main__":
    main()
