#!/usr/bin/env python3
"""What does the split-f16 K loop's STRUCTURE sustain, with no epilogue at all?  (tools/simd_overlap.hip; docs/EXPERIMENTS.md 3.2)
One MFMA wave per SIMD (36 MFMAs + 20 fragment reads per K tile = the conv's wave tile) beside one loader wave per SIMD (10 pieces of 1 KiB per
K tile = the conv's operand stream), each configuration looped for 3 s (the numbers are sustained ones, taken from the second half), with the
pieces coming from a 2 MiB window (every piece an L2 hit) or a 64 MiB one (every piece an L2 miss served by the MALL).  Prints one JSON line each.

    python tools/kloop_ceiling.py
"""
import ctypes as C, json, os, subprocess, sys, time, torch, re
HERE = os.path.dirname(os.path.abspath(__file__))
out_dir = os.path.join(os.path.dirname(HERE), "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
so = os.path.join(out_dir, "libsimd_overlap.so")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", os.path.join(HERE, "simd_overlap.hip"), "-o", so], check=True)
lib = C.CDLL(so)
lib.overlap_launch.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
dev = torch.device("cuda:0")
cus = torch.cuda.get_device_properties(dev).multi_processor_count
A = (torch.relu(torch.randn(cus * 4 * 64 * 10, 8, device=dev)) * 0.5).half().contiguous()
out = torch.empty(cus * 512, device=dev)
iters = 20000
def smi():
    try:
        o = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
        card = next(iter(json.loads(o).values()))
        w = next((float(v) for k, v in card.items() if "ower" in k and re.match(r"^[0-9.]+$", str(v))), None)
        sclk = next((float(re.sub(r"[^0-9.]", "", str(v))) for k, v in card.items() if k.lower().startswith("sclk") and re.search(r"[0-9]", str(v))), None)
        return w, sclk
    except Exception:
        return None, None
def sustained(load, reads, pieces, do_compute, window, src, secs=3.0):
    st = torch.cuda.current_stream().cuda_stream
    t_end = time.time() + secs
    times, ws, cl = [], [], []
    while time.time() < t_end:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): lib.overlap_launch(load, reads, A.data_ptr(), src.data_ptr(), window, iters, pieces, do_compute, out.data_ptr(), cus, st, 0)
        e1.record()
        w, s = smi()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) * 1e3 / (4 * iters)); ws.append(w); cl.append(s)
        cyc = float(out.view(cus, 512)[:, 0].mean()) / iters
    h = len(times) // 2
    us = sum(times[h:]) / len(times[h:])
    return {"us_per_iter": round(us, 4), "cycles": round(cyc), "clock_from_cycles_ghz": round(cyc / us / 1e3, 3), "power_w": ws[-1], "sclk": cl[-1], "first_us": round(times[0], 4),
            "mfma_tflops": round(cus * 4 * 36 * 32768 / us / 1e6) if do_compute else None}
for wmb in (2, 64):
    window = wmb << 20
    src = torch.randint(0, 255, (window + (1 << 20),), dtype=torch.uint8, device=dev)
    print(json.dumps({"window_mb": wmb, "what": "compute only (1 MFMA wave/SIMD, fragment reads)", **sustained(0, 1, 0, 1, window, src)}), flush=True)
    print(json.dumps({"window_mb": wmb, "what": "compute + loader 10 pieces sc1", **sustained(2, 1, 10, 1, window, src)}), flush=True)
    print(json.dumps({"window_mb": wmb, "what": "loader only", **sustained(2, 1, 10, 0, window, src)}), flush=True)
    del src
