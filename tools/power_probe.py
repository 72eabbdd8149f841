#!/usr/bin/env python3
"""Is a hidden-conv chain kernel power-bound?  Loops the sampler's chained launch (ehm_gcn_hidden_stack at the benchmark shape) for a few
seconds while sampling the socket power and the shader clock the SMU reports (rocm-smi), for relu-like activations (what the sampler
feeds the kernel) and for all-zero operands (the same instruction stream with nothing toggling).

    python tools/power_probe.py f16x3|f16 [seconds] [B]        -> one JSON line per data variant
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

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egohmr_amd import _lib  # noqa: E402
from egohmr_amd.factory import build_synthetic_model  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda:0")
model = build_synthetic_model(dev, 0)
model.gcn_precision = prec
L = _lib.lib()
h = model.fused_sampler.gcn()
hid, tile = model.diffusion_model.hid_dim, L.ehm_gcn_row_tile()
rows_pad = (2 * B * 24 + tile - 1) // tile * tile
nl = 2 * model.diffusion_model.num_layers
flops = 2 * B * (24 * 2 * hid * hid + 24 * 24 * hid) * 2.0


def smi():
    """(watts, sclk MHz) from rocm-smi; None where it cannot be parsed."""
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out)
        card = next(iter(d.values()))
        w = next((float(v) for k, v in card.items() if "ower" in k and re.match(r"^[0-9.]+$", str(v))), None)
        sclk = next((float(re.sub(r"[^0-9.]", "", str(v))) for k, v in card.items() if k.lower().startswith("sclk") and re.search(r"[0-9]", str(v))), None)
        return w, sclk
    except Exception:
        return None, None


for variant in ("relu_like", "zeros", "dense_random"):
    g = torch.Generator(device=dev).manual_seed(1)
    X = torch.randn(rows_pad, hid, device=dev, generator=g)
    X = torch.relu(X) * 0.5 if variant == "relu_like" else X * 0.0 if variant == "zeros" else X
    X2, Y1, Y2 = torch.empty_like(X), torch.empty_like(X), torch.empty_like(X)
    _lib.check(L.ehm_gcn_pack_activations(X.data_ptr(), X2.data_ptr(), rows_pad, hid, L.ehm_gcn_activation_group(h), None))
    X0 = X2.clone()
    bufs = (C.c_void_p * 3)(X2.data_ptr(), Y1.data_ptr(), Y2.data_ptr())
    res = C.c_int(0)
    stop, count = threading.Event(), [0]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def work():
        torch.cuda.set_device(dev)
        e0.record()
        while not stop.is_set():
            for _ in range(20):
                if variant != "zeros":
                    X2.copy_(X0)              # every chain starts from the same activations
                _lib.check(L.ehm_gcn_hidden_stack(h, bufs, rows_pad, C.byref(res), None))
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
    ws = [w for w, _ in samples if w is not None]
    cs = [c for _, c in samples if c is not None]
    # per-conv time incl. the activation restore copy of the non-zero variants (25-50 MB per 8 convs: < 2 %)
    print(json.dumps({"precision": prec, "data": variant, "us_per_conv_incl_restore_copy": ms * 1e3 / (count[0] * nl),
                      "algorithmic_tflops": flops / (ms * 1e-3 / (count[0] * nl)) / 1e12, "socket_power_w_avg": sum(ws) / len(ws) if ws else None,
                      "socket_power_w_max": max(ws) if ws else None, "sclk_mhz_avg": sum(cs) / len(cs) if cs else None, "smi_samples": len(samples)}), flush=True)
