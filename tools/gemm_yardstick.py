#!/usr/bin/env python3
"""The vendor library on the hidden conv's GEMM: what does hipBLASLt (through torch.matmul, f16 inputs, f32 accumulate) sustain on
[rows = 12288, K = 1024] x [1024, 2 x 1024] - the two branches of one f16 hidden conv at B = 256 x 2 passes - on relu-like activations, with NO
epilogue (no adjacency mix, BN, ReLU, residual, no f16 conversion of the result)?  The yardstick for gcn_hidden_chain_kernel<1, 8> (docs/EXPERIMENTS.md 3.2).

    python tools/gemm_yardstick.py [seconds]
"""
import json
import sys
import time

import torch

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for rows, K, N in ((12288, 1024, 2048), (12288, 1024, 1024), (61440, 1024, 2048)):
    for data in ("relu_like", "dense_random", "zeros"):
        x = torch.randn(rows, K, device=dev, generator=g)
        x = {"relu_like": torch.relu(x) * 0.5, "dense_random": x, "zeros": torch.zeros_like(x)}[data].half()
        w = (torch.randn(K, N, device=dev, generator=g) * 0.03).half()
        if data == "zeros":
            w.zero_()
        y = torch.empty(rows, N, device=dev, dtype=torch.half)
        for _ in range(20):
            torch.matmul(x, w, out=y)
        torch.cuda.synchronize()
        t_end, n, spans = time.time() + secs, 0, []
        while time.time() < t_end:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                torch.matmul(x, w, out=y)
            e1.record()
            torch.cuda.synchronize()
            spans.append(e0.elapsed_time(e1) / 200)
        ms = sum(spans[len(spans) // 2:]) / len(spans[len(spans) // 2:])          # sustained: second half of the window
        print(json.dumps({"gemm": f"[{rows},{K}]x[{K},{N}] f16 -> f16, f32 accumulate (torch.matmul / hipBLASLt)", "data": data, "us": round(ms * 1e3, 2),
                          "tflops": round(2.0 * rows * K * N / (ms * 1e-3) / 1e12, 1), "frac_of_2500": round(2.0 * rows * K * N / (ms * 1e-3) / 2.5e15, 3)}), flush=True)
