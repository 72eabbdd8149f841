#!/usr/bin/env python3
"""The power-limited ceiling of the f16 matrix pipe on an MI355X, for operands that toggle like the hidden convs' - the yardstick for
`roofline.frac` of the two chained conv kernels, which both run at the 1400 W socket cap (tools/power_probe.py).

Builds tools/mfma_ceiling.hip with hipcc (into gpurun_out/), runs a register-only MFMA loop (8 waves per CU, 3 x 2 accumulators per wave, the
fragments cycling through 8 A and 4 B registers) for a few seconds per data variant while sampling rocm-smi, prints one JSON line each:

    python tools/mfma_ceiling.py [seconds]
"""
import ctypes as C
import json
import os
import re
import subprocess
import sys
import threading
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
out_dir = os.path.join(REPO, "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
so = os.path.join(out_dir, "libmfma_ceiling.so")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", os.path.join(HERE, "mfma_ceiling.hip"), "-o", so], check=True)
lib = C.CDLL(so)
lib.mfma_ceiling_launch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
blocks = torch.cuda.get_device_properties(dev).multi_processor_count
n = blocks * 8 * 64


def smi():
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
        card = next(iter(json.loads(out).values()))
        w = next((float(v) for k, v in card.items() if "ower" in k and re.match(r"^[0-9.]+$", str(v))), None)
        sclk = next((float(re.sub(r"[^0-9.]", "", str(v))) for k, v in card.items() if k.lower().startswith("sclk") and re.search(r"[0-9]", str(v))), None)
        return w, sclk
    except Exception:
        return None, None


def split(x):
    hi = x.half()
    lo = (x - hi.float()).half()
    return hi, lo


g = torch.Generator(device=dev).manual_seed(3)
cases = []
xa, xb = torch.randn(8, n, 8, device=dev, generator=g), torch.randn(4, n, 8, device=dev, generator=g) * 0.03
cases.append(("f16_dense_random", 0, xa.half(), xb.half()))
cases.append(("f16_relu_like_activations", 0, (torch.relu(xa) * 0.5).half(), xb.half()))
cases.append(("f16_zeros", 0, torch.zeros_like(xa).half(), torch.zeros_like(xb).half()))
ah, al = split(torch.relu(xa[:4]) * 0.5)
bh, bl = split(xb[:2])
cases.append(("f16x3_relu_like_activations", 1, torch.cat([ah, al]), torch.cat([bh, bl])))
ah, al = split(xa[:4])
cases.append(("f16x3_dense_random", 1, torch.cat([ah, al]), torch.cat([bh, bl])))
for name, variant, A, B in list(cases):                     # the same operands through v_mfma_f32_16x16x32_f16 (variants 2 / 3)
    if "zeros" not in name:
        cases.append((name + "_mfma16x16x32", variant + 2, A, B))
out = torch.empty(n, device=dev)
iters = 4000
for name, variant, A, B in cases:
    A, B = A.contiguous(), B.contiguous()
    per_iter = {0: 24, 1: 72, 2: 48, 3: 144}[variant]
    flop_per_mfma = (32 * 32 * 16 * 2) if variant < 2 else (16 * 16 * 32 * 2)
    stop, count = threading.Event(), [0]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def work():
        torch.cuda.set_device(dev)
        e0.record()
        while not stop.is_set():
            for _ in range(4):
                rc = lib.mfma_ceiling_launch(variant, A.data_ptr(), B.data_ptr(), out.data_ptr(), blocks, iters, torch.cuda.current_stream().cuda_stream)
                assert rc == 0, rc
                count[0] += 1
            torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()

    th = threading.Thread(target=work)
    th.start()
    time.sleep(0.8)
    samples, t0 = [], time.time()
    while time.time() - t0 < secs:
        samples.append(smi())
        time.sleep(0.2)
    stop.set()
    th.join()
    ms = e0.elapsed_time(e1)
    flops = count[0] * float(iters) * per_iter * blocks * 8 * flop_per_mfma
    ws = [w for w, _ in samples if w is not None]
    cs = [c for _, c in samples if c is not None]
    print(json.dumps({"case": name, "issued_mfma_tflops": flops / (ms * 1e-3) / 1e12, "frac_of_2500": flops / (ms * 1e-3) / 2.5e15,
                      "socket_power_w_avg": sum(ws) / len(ws) if ws else None, "sclk_mhz_avg": sum(cs) / len(cs) if cs else None,
                      "finite": bool(torch.isfinite(out).all())}), flush=True)
