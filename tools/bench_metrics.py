#!/usr/bin/env python3
"""Timing of the evaluation block's kernels at the driver's shapes (test_egohmr.py:399-505): the contact score's nearest-neighbour
search (brute force through LDS tiles, against the f32 vector rate), V2V / MPJPE / PA-MPJPE / diversity.   python tools/bench_metrics.py [B] [S] [N]  -> one JSON line"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egohmr_amd import metrics  # noqa: E402

F32_VECTOR_PEAK = 157.3e12          # MI355X, packed FP32 FMA (MI355X_MICROARCH.md)


def measure(B=128, S=10, N=20000, reps=5, dev=None):
    dev = dev or torch.device("cuda:0")
    g = np.random.Generator(np.random.PCG64(3))
    nb = B * S
    # a room of surfaces (walls / floor), bodies standing in it: the contact score's geometry
    room = np.array([5.0, 3.0, 4.0])
    y = g.uniform(0, 1, (B, N, 3)) * room
    face = g.integers(0, 3, (B, N))
    for a in range(3):
        y[..., a] = np.where(face == a, np.round(y[..., a] / room[a]) * room[a], y[..., a])
    center = np.array([2.5, 0.9, 2.0])
    x = center + np.array([0.25, 0.5, 0.2]) * g.normal(size=(nb, 6890, 3))
    yd = torch.from_numpy(y.astype(np.float32)).to(dev).unsqueeze(1).expand(-1, S, -1, -1).reshape(nb, N, 3).contiguous()
    xd = torch.from_numpy(x.astype(np.float32)).to(dev)

    def timed(fn, n=reps):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    out = {"bodies": nb, "scene_points": N}
    t_nn = timed(lambda: metrics.contact_score(xd, yd), n=3)
    pairs = nb * 6890 * N
    out["contact_score"] = {"ms": t_nn, "bodies_per_s": nb / t_nn * 1e3, "kernel": "nn_dist2_kernel (brute force through LDS tiles)", "tflops_8_flop_per_pair": 8 * pairs / t_nn / 1e9,
                            "frac_of_f32_vector_peak": 8 * pairs / (t_nn * 1e-3) / F32_VECTOR_PEAK, "bound": "f32 vector ALU (157.3 TFLOP/s)"}
    pj, gj = torch.randn(B, S, 24, 3, device=dev), torch.randn(B, 24, 3, device=dev)
    pv, gv = torch.randn(B, S, 6890, 3, device=dev), torch.randn(B, 6890, 3, device=dev)
    jm, vm = torch.rand(B, 24, device=dev) < 0.6, torch.rand(B, 6890, device=dev) < 0.6
    t = timed(lambda: metrics.point_errors(pv, gv, mask=vm))
    out["v2v"] = {"ms": t, "GB_per_s": (pv.numel() + gv.numel()) * 4 / t / 1e6}
    out["mpjpe_ms"] = timed(lambda: metrics.point_errors(pj, gj, mask=jm))
    out["pa_mpjpe_ms"] = timed(lambda: metrics.procrustes(pj, gj, mask=jm))
    out["diversity_ms"] = timed(lambda: metrics.diversity(pj, jm))
    return out


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:4]] + [128, 10, 20000][len(sys.argv) - 1:]
    print(json.dumps(measure(a[0], a[1], a[2])))
