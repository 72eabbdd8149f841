#!/usr/bin/env python3
"""Per-CU operand-ingest ceiling of an MI355X from L2: bytes per clock through global_load_lds (plain / sc1), through global_load_dwordx4
into registers, and through both alternating - with nothing else running (tools/ingest_ceiling.hip).  The chained conv kernels stage
80 KiB (split-f16) / 56 KiB (f16) per K tile and CU and are measured at 24-26 B/clk (docs/EXPERIMENTS.md 3.2); this is what the path can do alone.

    python tools/ingest_ceiling.py            -> one JSON line per mode
"""
import ctypes as C
import json
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
out_dir = os.path.join(os.path.dirname(HERE), "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
so = os.path.join(out_dir, "libingest_ceiling.so")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", os.path.join(HERE, "ingest_ceiling.hip"), "-o", so], check=True)
lib = C.CDLL(so)
lib.ingest_launch.argtypes = [C.c_int, C.c_void_p, C.c_uint, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
cus = torch.cuda.get_device_properties(dev).multi_processor_count
window = 2 << 20
src = torch.randint(0, 255, (window + 65536,), dtype=torch.uint8, device=dev)
sink = torch.zeros(4, dtype=torch.int32, device=dev)
iters = 4000
for mode, name in ((0, "global_load_lds"), (1, "global_load_lds sc1"), (2, "global_load_dwordx4 -> VGPR"), (3, "alternating lds / VGPR")):
    for blocks in (cus, 2 * cus):
        if blocks > cus and True:
            pass
        st = torch.cuda.current_stream().cuda_stream
        assert lib.ingest_launch(mode, src.data_ptr(), window, 200, sink.data_ptr(), blocks, st) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        assert lib.ingest_launch(mode, src.data_ptr(), window, iters, sink.data_ptr(), blocks, st) == 0
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        nbytes = blocks * 8 * iters * 8 * 1024
        print(json.dumps({"mode": name, "blocks": blocks, "waves_per_cu": 8 * blocks // cus, "TB_per_s": nbytes / (ms * 1e-3) / 1e12,
                          "bytes_per_clk_per_cu_at_2.4GHz": nbytes / (ms * 1e-3) / cus / 2.4e9}), flush=True)
        if blocks == cus and 8 * 8 * 1024 * 2 > 160 * 1024:
            break
