#!/usr/bin/env python3
"""Socket power and SMU shader clock while the PRODUCTION sampling loop runs (the same call bench.py times: B = 256, 100-step DDPM, every
step split-f16, encoders inside the call) - the in-situ operating point of `gcn_hidden_chain_kernel<3, 4>`, not the tools/bench_hidden.py harness.

    python tools/power_probe_bench.py [seconds]        -> one JSON line
"""
import json
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egohmr_amd.synthetic as syn  # noqa: E402
from egohmr_amd.diffusion import create_gaussian_diffusion  # noqa: E402
from egohmr_amd.factory import batch_to_device, build_synthetic_model  # noqa: E402


def smi():
    """(watts, sclk MHz) from rocm-smi; None where it cannot be parsed."""
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
        card = next(iter(json.loads(out).values()))
        w = next((float(v) for k, v in card.items() if "ower" in k and re.match(r"^[0-9.]+$", str(v))), None)
        sclk = next((float(re.sub(r"[^0-9.]", "", str(v))) for k, v in card.items() if k.lower().startswith("sclk") and re.search(r"[0-9]", str(v))), None)
        return w, sclk
    except Exception:
        return None, None


secs = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
B, N, n = 256, 4096, 100
dev = torch.device("cuda:0")
model = build_synthetic_model(dev, 0, diffuse_fuse=True, sensitive=dict(num_diffusion_timesteps=n))
model.gcn_precision = "f16x3"
model.f16x3_last_steps = n
diffusion = create_gaussian_diffusion(num_diffusion_timesteps=n, timestep_respacing="")
batch = batch_to_device(syn.make_batch(B, N, seed=100), dev)
noise = torch.from_numpy(syn.make_noise_stack(diffusion.num_timesteps, B, seed=100)).to(dev)
fs = model.fused_sampler


def call():
    fs.invalidate()
    fs.run_samples(diffusion, batch, [noise], ddim=False, guided=False, cond_grad_weight=1.0, defer_status=True)


call()
torch.cuda.synchronize()
stop, count = threading.Event(), [0]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def work():
    torch.cuda.set_device(dev)
    e0.record()
    while not stop.is_set():
        call()
        count[0] += 1
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()


th = threading.Thread(target=work)
th.start()
time.sleep(1.0)
samples, t0 = [], time.time()
while time.time() - t0 < secs:
    samples.append(smi())
    time.sleep(0.2)
stop.set()
th.join()
fs.check_status()
ms = e0.elapsed_time(e1) / max(count[0], 1)
ws = [w for w, _ in samples if w is not None]
cs = [c for _, c in samples if c is not None]
print(json.dumps({"what": "production sampling call looped (B256, DDPM-100, all steps split-f16, encoders in the call)", "ms_per_call": ms,
                  "bodies_per_s": B / ms * 1e3, "calls": count[0], "socket_power_w_avg": sum(ws) / len(ws) if ws else None,
                  "socket_power_w_max": max(ws) if ws else None, "sclk_mhz_avg": sum(cs) / len(cs) if cs else None,
                  "sclk_mhz_min": min(cs) if cs else None, "sclk_mhz_max": max(cs) if cs else None, "smi_samples": len(samples)}), flush=True)
