#!/usr/bin/env python3
"""Can the second wave of a SIMD stage operands while the first one issues MFMAs?  (tools/simd_overlap.hip; docs/EXPERIMENTS.md 3.2)
Per iteration: compute waves 36 MFMAs (= one split-f16 K tile of a wave, 1152 matrix cycles per SIMD), loader waves P pieces of 1 KiB.
Prints cycles per iteration (at the SMU's clock, whatever it is: ratios are what matters) for compute alone, loader alone, both.

    python tools/simd_overlap.py
"""
import ctypes as C
import json
import os
import subprocess

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
out_dir = os.path.join(os.path.dirname(HERE), "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
so = os.path.join(out_dir, "libsimd_overlap.so")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", os.path.join(HERE, "simd_overlap.hip"), "-o", so], check=True)
lib = C.CDLL(so)
lib.overlap_launch.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
dev = torch.device("cuda:0")
cus = torch.cuda.get_device_properties(dev).multi_processor_count
window = 2 << 20
src = torch.randint(0, 255, (window + (1 << 20),), dtype=torch.uint8, device=dev)
A = (torch.relu(torch.randn(cus * 4 * 64 * 10, 8, device=dev)) * 0.5).half().contiguous()
out = torch.empty(cus * 512, device=dev)
iters = 3000
last_cycles = 0.0


def run(load, reads, pieces, do_compute, valu=0):
    st = torch.cuda.current_stream().cuda_stream
    assert lib.overlap_launch(load, reads, A.data_ptr(), src.data_ptr(), window, 100, pieces, do_compute, out.data_ptr(), cus, st, valu) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert lib.overlap_launch(load, reads, A.data_ptr(), src.data_ptr(), window, iters, pieces, do_compute, out.data_ptr(), cus, st, valu) == 0
    e1.record()
    torch.cuda.synchronize()
    global last_cycles
    last_cycles = float(out.view(cus, 512)[:, 0].mean()) / iters       # shader cycles per iteration (s_memtime of each block's wave 0)
    return e0.elapsed_time(e1) * 1e3 / iters        # us per iteration


names = {1: "global_load_lds", 2: "global_load_lds sc1", 3: "global_load_dwordx4 -> VGPR"}
for reads in (0, 1):
    base = run(0, reads, 0, 1)
    print(json.dumps({"compute_only_us_per_iter": base, "fragment_reads": bool(reads), "mfma_tflops": cus * 4 * 36 * 32768 / base / 1e6}), flush=True)
    for load in (1, 2, 3):
        for pieces in (10, 20):
            alone = run(load, reads, pieces, 0)
            both = run(load, reads, pieces, 1)
            print(json.dumps({"loader": names[load], "pieces_per_iter_per_wave": pieces, "fragment_reads": bool(reads), "loader_only_us": alone, "both_us": both,
                              "compute_only_us": base, "both_over_max": both / max(alone, base), "both_over_sum": both / (alone + base)}), flush=True)

# the helper wave of the engine: 10 pieces + a slice of the previous tile's epilogue (VALU) per K tile
for valu in (256, -256, 512, -512, 1024, -1024):
    alone = run(1, 1, 10, 0, valu); c_alone = last_cycles
    both = run(1, 1, 10, 1, valu); c_both = last_cycles
    base = run(0, 1, 0, 1); c_base = last_cycles
    print(json.dumps({"loader": "global_load_lds + VALU", "valu_fma_per_iter": valu, "cycles_loader_only": c_alone, "cycles_both": c_both, "cycles_compute_only": c_base,
                      "clock_ghz_both": c_both / both / 1e3, "clock_ghz_compute_only": c_base / base / 1e3, "pieces_per_iter_per_wave": 10, "fragment_reads": True, "loader_only_us": alone,
                      "both_us": both, "compute_only_us": base, "both_over_max": both / max(alone, base), "mfma_tflops_both": cus * 4 * 36 * 32768 / both / 1e6}), flush=True)

# pairwise: MFMA (+ fragment reads) beside a VALU-only partner
for reads in (0, 1):
    for valu in (512, 2048):
        alone = run(0, reads, 0, 0, valu); c_alone = last_cycles
        both = run(0, reads, 0, 1, valu); c_both = last_cycles
        base = run(0, reads, 0, 1); c_base = last_cycles
        print(json.dumps({"loader": "VALU only", "valu_fma_per_iter": valu, "fragment_reads": bool(reads), "cycles_loader_only": c_alone, "cycles_both": c_both,
                          "cycles_compute_only": c_base}), flush=True)

# the same pairs with the partner's vector-ALU work as PLAIN v_fmac_f32 (what the conv epilogue's adjacency mix is: 1128 v_fmac_f32 per lane and tile);
# the rows above use what hipcc makes of independent fmaf chains: v_pk_fma_f32, which MI355X_MICROARCH.md lists as an anti-lever beside MFMAs
lib.overlap_launch_plain_valu.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]


def run_plain(load, pieces, do_compute, valu):
    st = torch.cuda.current_stream().cuda_stream
    assert lib.overlap_launch_plain_valu(load, A.data_ptr(), src.data_ptr(), window, 100, pieces, do_compute, out.data_ptr(), cus, st, valu) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert lib.overlap_launch_plain_valu(load, A.data_ptr(), src.data_ptr(), window, iters, pieces, do_compute, out.data_ptr(), cus, st, valu) == 0
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters, float(out.view(cus, 512)[:, 0].mean()) / iters


for valu in (256, 512, 1024, 2048):
    _, c_alone = run_plain(0, 0, 0, valu)
    us_both, c_both = run_plain(0, 0, 1, valu)
    _, c_base = run_plain(0, 0, 1, 0)
    print(json.dumps({"loader": "VALU only, plain v_fmac_f32", "valu_fma_per_iter": valu, "fragment_reads": True, "cycles_loader_only": c_alone, "cycles_both": c_both,
                      "cycles_compute_only": c_base, "both_over_sum": c_both / (c_alone + c_base), "both_over_max": c_both / max(c_alone, c_base)}), flush=True)
for valu in (256, 512):
    _, c_alone = run_plain(1, 10, 0, valu)
    us_both, c_both = run_plain(1, 10, 1, valu)
    _, c_base = run_plain(0, 0, 1, 0)
    print(json.dumps({"loader": "global_load_lds + plain v_fmac_f32", "valu_fma_per_iter": valu, "pieces_per_iter_per_wave": 10, "fragment_reads": True, "cycles_loader_only": c_alone,
                      "cycles_both": c_both, "cycles_compute_only": c_base, "both_over_sum": c_both / (c_alone + c_base), "both_over_max": c_both / max(c_alone, c_base),
                      "mfma_tflops_both": cus * 4 * 36 * 32768 / us_both / 1e6}), flush=True)
