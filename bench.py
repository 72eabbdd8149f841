#!/usr/bin/env python3
"""Benchmark of the EgoHMR stage-2 sampling hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one whole sampling call over one batch of synthetic items already resident in HBM:
conditioning encoders (ResNet-50, scene PointNet, heads) once, then T denoising steps (Modulated-GCN,
two passes with diffuse_fuse, rot6d->rotmat + SMPL LBS in every step, sampler update), final decode, and
at N > 1 the single RCCL all-gather of the packed SMPL parameters.  The default workload is the one
BASELINE.json's metric is quoted on: 100-step DDPM, batch 256 per GPU, 1 sample/item, 4096 scene points.
Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

WORKLOADS = {
    # name: (num_diffusion_timesteps, respacing, description)
    "ddpm100": (100, "", "B256 S1 DDPM-100 N4096 ResNet50+PointNet cond, diffuse_fuse(2 GCN passes), LBS every step, unguided"),
    "c2_ddim10": (100, "ddim10", "BASELINE config 2: B256 S1 DDIM-10 N4096 ResNet50+PointNet cond, diffuse_fuse, LBS every step"),
    "c1_ddim5": (50, "ddim5", "BASELINE config 1 shape: DDIM-5 of 50"),
    # BASELINE config 3: B128 items x 10 samples, full 100-step DDPM, collision guidance on the last 11 steps (proxy loss);
    # a "step" = one batch of items = its 10 guided samples, run as ONE fused loop over 1280 bodies on ONE conditioning pass (the reference
    # runs 10 sequential loops and re-encodes in every step of each)
    "c3_guided": (100, "", "BASELINE config 3: B128 x S10 DDPM-100, collision-guided (last 11 steps), conditioning encoded once per item"),
}
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32-input MFMA peak
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak (~2.5 PFLOP/s)


def hidden_layer_flops(virtual_bodies: int, hid: int) -> float:
    """per body-pass and hidden conv 24*2*hid^2 MAC (W0,W1) + 24*24*hid MAC (adjacency mix)."""
    return virtual_bodies * (24 * 2 * hid * hid + 24 * 24 * hid) * 2.0


def kernel_source_sha1():
    import hashlib
    with open(os.path.join(REPO, "egohmr_amd", "csrc", "gcn_tile.hip"), "rb") as f:
        return hashlib.sha1(f.read()).hexdigest()


def pmc_traffic(precision):
    """HBM-side bytes per hidden conv of the chained launch, from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 gfx950
    correction + WRITE_SIZE, tools/pmc_round.sh -> profiles/pmc_traffic.json).  Counters cannot be read from inside this
    process, so the figure is only reported when it was collected for THIS kernel source (sha1 of csrc/gcn_tile.hip recorded next
    to it); otherwise None."""
    try:
        with open(os.path.join(REPO, "profiles", "pmc_traffic.json")) as f:
            e = json.load(f).get(precision, {})
    except OSError:
        return None
    return e.get("bytes_per_conv") if e.get("kernel_source_sha1") == kernel_source_sha1() else None


def time_dominant_kernel(model, B, passes, reps=5):
    """Average launch duration of the hidden Modulated-GCN conv at the benchmark's shape, HIP events on the launch stream.
    The convs are chained exactly as in the denoiser (block: Y1 = conv(X); X' = conv(Y1) + X), starting from a post-ReLU-like
    matrix, so the operands are real activations: the chip's clock under MFMA load depends on the data (random dense inputs
    run ~8 % slower than the sampler's half-zero activations), and the rocprofv3 average of the sampling loop is the
    number this has to agree with."""
    from egohmr_amd import _lib
    L = _lib.lib()
    hid = model.diffusion_model.hid_dim
    tile = L.ehm_gcn_row_tile()
    rows = passes * B * 24
    rows_pad = (rows + tile - 1) // tile * tile
    g = torch.Generator(device=model.device).manual_seed(1)
    X = torch.relu(torch.randn(rows_pad, hid, device=model.device, generator=g)) * 0.5
    Y1, Y2 = torch.empty_like(X), torch.empty_like(X)
    h = model.fused_sampler.gcn()
    if model.gcn_precision != "f32":     # split-f16 modes exchange activations in the X2 format
        X2 = torch.empty_like(X)
        _lib.check(L.ehm_gcn_pack_activations(X.data_ptr(), X2.data_ptr(), rows_pad, hid, L.ehm_gcn_activation_group(h), _lib.stream_ptr()))
        X = X2
    nl = 2 * model.diffusion_model.num_layers
    s = _lib.stream_ptr()

    X0 = X.clone()
    import ctypes as C
    bufs = (C.c_void_p * 3)(X.data_ptr(), Y1.data_ptr(), Y2.data_ptr())
    res = C.c_int(0)

    def sweep():    # the sampler's own call: all hidden convs of one GCN forward (one chained launch on the split-f16 path)
        _lib.check(L.ehm_gcn_hidden_stack(h, bufs, rows_pad, C.byref(res), s))

    sweep()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in ev:
        X.copy_(X0)          # every sweep starts from the same activations (outside the timed span)
        e0.record()
        sweep()
        e1.record()
    torch.cuda.synchronize()
    return sum(e0.elapsed_time(e1) for e0, e1 in ev) * 1e-3 / (reps * nl), rows_pad


def cpu_baseline(n, rs, num_scene_points, budget_s, faithful=True):
    """The oracle on a bounded sample: B=4 items, as many leading steps of the same schedule as fit the time budget.
    faithful=True: like the reference (encoders inside every step, two GCN passes, eager torch-CPU); faithful=False: the same CPU
    arithmetic with the step-invariant encoders hoisted out of the loop - separates what hoisting buys from what the GPU buys."""
    from egohmr_amd import synthetic as syn
    from oracle import model as om, schedule as osched
    # eager torch-CPU on small batches is fastest at 16 threads on the MI355X host (measured 16/32/64: 0.28/0.20/0.10 bodies/s); use what helps and report it
    cores = int(os.environ.get("EGOHMR_CPU_THREADS", min(os.cpu_count() or 1, 16)))
    torch.set_num_threads(cores)
    B = 4
    sd, asset = syn.make_state_dict(0), syn.make_smpl_asset(0)
    mean, std = syn.make_body_rep_stats(0)
    ref = om.EgoHMROracle(sd, asset, mean, std, faithful=faithful)
    bnp = syn.make_batch(B, num_scene_points, seed=0)
    batch = {k: ({kk: torch.from_numpy(vv) for kk, vv in v.items()} if isinstance(v, dict) else torch.from_numpy(v)) for k, v in bnp.items()}
    tab = osched.make_tables(n, rs)
    T = tab.num_timesteps
    noise = torch.from_numpy(syn.make_noise_stack(T, B, seed=0))
    x = noise[0]
    done, t0 = 0, time.perf_counter()
    with torch.no_grad():
        for k, i in enumerate(range(T - 1, -1, -1)):
            batch["x_t"] = x
            mo = ref(batch, torch.full((B,), tab.timestep_map[i], dtype=torch.long))
            x = float(np.float32(tab.posterior_mean_coef1[i])) * mo["pred_x_start"] + float(np.float32(tab.posterior_mean_coef2[i])) * x \
                + (0.0 if i == 0 else 1.0) * float(np.exp(0.5 * np.float32(tab.posterior_log_variance_clipped[i]))) * noise[1 + k]
            done += 1
            if faithful and time.perf_counter() - t0 > budget_s and done >= 2:
                break
    dt = time.perf_counter() - t0
    if faithful:
        per_call = dt / done * T
        how = "reference-faithful: ResNet-50 + PointNet re-run every step"
    else:       # the first step carries the one-off encoders: time it apart
        per_call = dt if done == T else float("nan")
        how = "encoders hoisted (run once), otherwise the same eager CPU arithmetic"
    return {"value": B / per_call, "unit": "bodies/s", "cores": cores, "kind": "port",
            "sample": f"oracle/ (CPU restatement, {how}, 2 GCN passes, eager torch-CPU fp32, {cores} threads): B={B} items, "
                      f"{done} of {T} steps timed ({dt:.1f} s)" + (f", extrapolated linearly to {T} steps" if faithful and done < T else "")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="ddpm100", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=256, help="items per GPU")
    ap.add_argument("--scene-points", type=int, default=4096)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="time budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--no-lbs-every-step", action="store_true")
    ap.add_argument("--precision", default=os.environ.get("EGOHMR_GCN_PRECISION", "f16x3"), choices=["f32", "f16x3", "f16"],
                    help="arithmetic of the hidden GCN convs (DESIGN.md 3.2): f32 MFMA | split-f16 MFMA (f32-grade) | plain f16 (not parity-grade)")
    args = ap.parse_args()

    from egohmr_amd import dist as edist
    from egohmr_amd import synthetic as syn
    from egohmr_amd.diffusion import create_gaussian_diffusion
    from egohmr_amd.factory import batch_to_device, build_synthetic_model

    rank, world, local = edist.init_from_env()
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a HIP device; there is no CPU fallback"
    local %= torch.cuda.device_count()          # (functional 2-rank runs on a 1-GPU box share the device)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    n, rs, desc = WORKLOADS[args.workload]
    B, N = args.batch, args.scene_points
    S, guided = 1, False
    if args.workload == "c3_guided":
        S, guided = 10, True
        if args.batch == 256:
            B = 128
    model = build_synthetic_model(dev, 0, diffuse_fuse=True)
    model.lbs_every_step = not args.no_lbs_every_step
    model.gcn_precision = args.precision
    diffusion = create_gaussian_diffusion(num_diffusion_timesteps=n, timestep_respacing=rs)
    T = diffusion.num_timesteps
    batch = batch_to_device(syn.make_batch(B, N, seed=100 + rank), dev)            # inputs resident in HBM before timing
    noise = torch.from_numpy(syn.make_noise_stack(T, B, seed=100 + rank)).to(dev)
    noises = [noise] + [torch.from_numpy(syn.make_noise_stack(T, B, seed=100 + rank + 1000 * k)).to(dev) for k in range(1, S)]
    if guided:   # put the floor through the bodies so that the collision term is live (as in the guided golden)
        batch["scene_pcd_verts_full"][:, : N // 3, 1] = batch["smpl_params"]["transl"][:, None, 1] - 0.6
    fs = model.fused_sampler
    ddim = bool(rs)

    def one_step():
        fs.invalidate()                                                              # conditioning is part of the job: re-encode
        # the S samples of an item share its conditioning and are independent given it: one fused loop over S*B bodies
        # (FusedSampler.run_samples; bit-equal to S sequential loops, tests/test_gpu_api.py)
        # defer_status: the chain-status word of this call is read when the next call starts (and by check_status() behind the timed
        # region) instead of with a host wait at the end of every call
        outs = fs.run_samples(diffusion, batch, noises[:S], ddim=ddim, guided=guided, cond_grad_weight=2.0 if guided else 1.0, defer_status=True)
        packs = [edist.pack_params(o["other_outputs"]["pred_smpl_params"]) for o in outs]
        return edist.gather_packed(torch.cat(packs, 0)), outs[-1]

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    edist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gathered, res = one_step()
    torch.cuda.synchronize()
    fs.check_status()                                                                # (inside the timed region: the last call's status word)
    edist.barrier()
    torch.cuda.synchronize()
    dt = edist.max_over_ranks(time.perf_counter() - t0, dev)
    assert torch.isfinite(res["other_outputs"]["pred_vertices"]).all()
    assert gathered.shape == (world * B * S, edist.PACKED_WIDTH)

    # comparison legs (rank-local, N=1 only): the same job, same noise, (a) WITHOUT the precision schedule (every step split-f16),
    # (b) with the hidden convs on the f32-input MFMA (exact f32 products), (c) on plain f16 operands in every step (the "fp16
    # denoiser" of BASELINE config 5; NOT parity-grade) - each with its distance to the default path's vertices / joints
    legs = {}
    if args.precision == "f16x3" and world == 1 and S == 1:
        ref_v = res["other_outputs"]["pred_vertices"].float().clone()
        ref_j = res["other_outputs"]["pred_keypoints_3d"].float().clone()

        def leg(prec, last_steps):
            old = (model.gcn_precision, model.f16x3_last_steps)
            model.gcn_precision, model.f16x3_last_steps = prec, last_steps
            try:
                one_step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                _, r = one_step()
                torch.cuda.synchronize()
                fs.check_status()
                d = time.perf_counter() - t1
            finally:
                model.gcn_precision, model.f16x3_last_steps = old
            v = r["other_outputs"]["pred_vertices"].float()
            dv = (v - ref_v).norm(dim=-1)                                  # per-vertex distance [B, V], metres
            return {"value": B / d, "unit": "bodies/s", "ms_per_step": d * 1e3,
                    "vs_default_path": {"max_vertex_dist_mm": float(dv.max()) * 1e3, "mean_v2v_mm": float(dv.mean()) * 1e3,
                                        "mpjpe_mm": float((r["other_outputs"]["pred_keypoints_3d"].float() - ref_j).norm(dim=-1).mean()) * 1e3}}

        legs["all_steps_f16x3"] = leg("f16x3", None)
        legs["f32_mfma_path"] = leg("f32", None)
        legs["f16_denoiser_path"] = leg("f16", None)
        legs["f16_denoiser_path"]["note"] = ("plain f16 operands and f16 activations in the hidden convs of EVERY step (f32 accumulate, everything "
                                             "else f32): BASELINE config 5's fp16 denoiser, not a parity path on its own")

    # split of one call (rank 0, informative)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    fs.invalidate()
    st = fs.prepare(batch)
    torch.cuda.synchronize()
    t_enc = time.perf_counter() - t1

    if rank == 0:
        passes = 2
        hid = model.diffusion_model.hid_dim
        flops = hidden_layer_flops(passes * B, hid)
        n_hidden = 2 * model.diffusion_model.num_layers
        lowprec = fs.lowprec_steps(T, min(T, 11) if guided else 0, ddim) if args.precision == "f16x3" else 0   # (the reference guides the last 11 steps: t <= 10)           # leading steps on plain f16 operands
        kernels = {}                                                                # live HIP-event timing of each conv kernel this job runs

        def time_kernel(prec):
            old = model.gcn_precision
            model.gcn_precision = prec
            try:
                kd, _ = time_dominant_kernel(model, B, passes)
            finally:
                model.gcn_precision = old
            peak = PEAK_F32_MFMA_TFLOPS if prec == "f32" else PEAK_F16_MFMA_TFLOPS
            per_prod = 3 if prec == "f16x3" else 1
            name = {"f32": "gcn_hidden_kernel (f32-input MFMA GEMM + fused modulated-adjacency/BN/ReLU epilogue), one launch per conv",
                    "f16x3": "gcn_hidden_chain_kernel<3, 4> (csrc/gcn_tile.hip): the 8 hidden convs of a GCN forward chained in one launch (4-wave 192x64 tiles); split-f16 operands, "
                             "3 MFMA per algorithmic product, f32 accumulate, fused modulated-adjacency/BN/ReLU/residual epilogue",
                    "f16": "gcn_hidden_chain_kernel<1, 8> (csrc/gcn_tile.hip): the 8 hidden convs chained in one launch (8-wave 192x128 tiles); plain f16 operands and f16 "
                           "activations, f32 accumulate, adjacency mix on the matrix cores, fused BN/ReLU/residual epilogue"}[prec]
            return {"bound": "mfma", "kernel": name, "achieved": flops / kd / 1e12, "peak": peak, "unit": "TFLOP/s", "frac": flops / kd / 1e12 / peak,
                    "mfma_flops_per_algorithmic_flop": per_prod, "issued_mfma_frac": flops * per_prod / kd / 1e12 / peak,
                    "traffic": pmc_traffic(prec), "avg_launch_ms": kd * 1e3, "avg_launch_ms_is": "per conv = chained launch / 8" if prec != "f32" else "per conv",
                    "flops_per_launch": flops, "formula": "virtual_bodies*(24*2*hid^2 + 24*24*hid)*2, virtual_bodies = passes*B (SURVEY 8d, hoisted)"}

        if args.precision == "f16x3":
            kernels["f16x3"] = time_kernel("f16x3")
            kernels["f16x3"]["launches_per_call"], kernels["f16x3"]["ms_per_call"] = T - lowprec, kernels["f16x3"]["avg_launch_ms"] * n_hidden * (T - lowprec)
            if lowprec:
                kernels["f16"] = time_kernel("f16")
                kernels["f16"]["launches_per_call"], kernels["f16"]["ms_per_call"] = lowprec, kernels["f16"]["avg_launch_ms"] * n_hidden * lowprec
        else:
            kernels[args.precision] = time_kernel(args.precision)
            kernels[args.precision]["launches_per_call"] = T
            kernels[args.precision]["ms_per_call"] = kernels[args.precision]["avg_launch_ms"] * n_hidden * T
        dominant = max(kernels, key=lambda k: kernels[k]["ms_per_call"])            # the kernel the job spends most time in
        value = world * B * S * args.steps / dt
        flops_per_body = 183.8e9 if args.workload == "ddpm100" else None            # SURVEY 8d, hoisted, T = 100 with diffuse_fuse
        out = {
            "metric": "sampled bodies/sec (100-step DDPM, batch 256)" if args.workload == "ddpm100" else f"sampled bodies/sec ({args.workload})",
            "value": value,
            "unit": "bodies/s",
            "n_gpus": world,
            "n_ranks_seen": world if world == 1 else int(torch.distributed.get_world_size()),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"f32": "f32",
                      "f16x3": "f32 results: denoiser GEMMs as 3x f16 MFMA on hi/lo-split operands (f32 accumulate) on the last "
                               f"{T - lowprec} of {T} steps, plain f16 operands on the first {lowprec} (precision schedule, DESIGN.md 3.6; "
                               "final bodies within 1e-5 m of the all-split run, measured below)",
                      "f16": "f16 denoiser GEMMs and activations (f32 accumulate) + f32 everything else"}[args.precision],
            "data": "synthetic",
            "config": {"workload": desc, "name": args.workload, "items_per_gpu": B, "samples_per_item": S, "samples_in_one_loop": bool(S > 1), "collision_guided": guided, "denoising_steps": T,
                       "scene_points": N, "gcn_passes_per_step": passes, "lbs_every_step": bool(model.lbs_every_step),
                       "gcn_precision": args.precision, "f16x3_last_steps": (T - lowprec) if args.precision == "f16x3" else None,
                       "f16x3_last_steps_policy": str(model.f16x3_last_steps),
                       "pass_pruning": {"items_without_second_pass": int(B - st.num_masked) if model.prune_passes else 0, "of": B,
                                        "note": "exact (egohmr.py:249-254): all-visible items skip the image-masked pass; the synthetic "
                                                "visibility draw (Bernoulli 0.6 per OpenPose joint, SURVEY 8d) almost never produces one"},
                       "weights": "seeded random (no checkpoint offline)", "smpl": "synthetic SMPL-shaped asset",
                       "parallelism": f"items sharded x{world}, one RCCL all-gather of [B,226] at the end"},
            "roofline": kernels[dominant],
            "roofline_other_kernel": {k: v for k, v in kernels.items() if k != dominant} or None,
            "breakdown_ms": {"encoders_and_projections_once": t_enc * 1e3, "per_call_total": dt / args.steps * 1e3,
                             "hidden_convs_est": sum(v["ms_per_call"] for v in kernels.values())},
            "target": {"north_star_bodies_per_s": 10000,
                       "parity_ceiling_bodies_per_s": (PEAK_F16_MFMA_TFLOPS / 3) * 1e12 / flops_per_body if flops_per_body else None,
                       "note": "183.8 GFLOP per body (SURVEY 8d); with every product as 3 MFMA the dense f16 peak allows 833 TFLOP/s algorithmic = the "
                               "ceiling above, so >= 10k bodies/s is out of reach at f32-grade arithmetic on every step"},
        }
        out.update(legs)
        if args.cpu_seconds > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(n, rs, N, args.cpu_seconds * 0.6)
            out["cpu_baseline_hoisted"] = cpu_baseline(n, rs, N, args.cpu_seconds * 0.4, faithful=False) if T <= 100 else None
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    edist.barrier()


if __name__ == "__main__":
    main()
