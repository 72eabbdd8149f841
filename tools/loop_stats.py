#!/usr/bin/env python3
"""Where the blocks of the one-launch sampling loop (csrc/gcn_tile.hip gcn_loop_kernel) spend their cycles.  Needs a library built with
EHM_HIPCC_FLAGS=-DEHM_LOOPSTAT (per-block s_memtime accounting):
    EHM_HIPCC_FLAGS=-DEHM_LOOPSTAT EHM_LIB_PATH=/tmp/libegohmr_stat.so python tools/loop_stats.py [ddim10|ddpm100]"""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egohmr_amd import _lib  # noqa: E402

_lib.build(force=not os.path.exists(_lib.LIB_PATH))
from egohmr_amd import synthetic as syn  # noqa: E402
from egohmr_amd.diffusion import create_gaussian_diffusion  # noqa: E402
from egohmr_amd.factory import batch_to_device, build_synthetic_model  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "ddim10"
dev = torch.device("cuda:0")
B = 256
model = build_synthetic_model(dev, 0, diffuse_fuse=True, sensitive=dict(num_diffusion_timesteps=100))
model.f16x3_last_steps = None
d = create_gaussian_diffusion(num_diffusion_timesteps=100, timestep_respacing="" if which == "ddpm100" else which)
T = d.num_timesteps
batch = batch_to_device(syn.make_batch(B, 4096, seed=100), dev)
noise = torch.from_numpy(syn.make_noise_stack(T, B, seed=100)).to(dev)
fs = model.fused_sampler
L = _lib.lib()
L.ehm_dbg_set_loopstat.argtypes = [C.c_void_p]
blocks = 2 * torch.cuda.get_device_properties(0).multi_processor_count
st = torch.zeros(blocks, 16, dtype=torch.int64, device=dev)
fs.run(d, batch, noise, ddim=bool(which != "ddpm100"))
torch.cuda.synchronize()
assert L.ehm_dbg_set_loopstat(st.data_ptr()) == 0
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
st.zero_()
fs.invalidate(); p = fs.prepare(batch); torch.cuda.synchronize()
t0.record(); fs.run(d, batch, noise, ddim=bool(which != "ddpm100"), prepared=p); t1.record()
torch.cuda.synchronize()
s = st.double().cpu()
is_item = s[:, 14] > 0
ti, it = s[~is_item], s[is_item]
tot = float(ti[:, 0].mean())
out = {"sampler": which, "loop_ms": t0.elapsed_time(t1), "tile_blocks": int((~is_item).sum()), "item_blocks": int(is_item.sum()),
       "tile_block": {"mean_cycles": tot, "wait_deps_frac": float(ti[:, 1].mean() / tot), "skip_fetch_frac": float(ti[:, 8].mean() / tot),
                      "tiles_per_block": float((ti[:, 10] + ti[:, 11]).mean()), "cycles_per_tile_incl_everything": tot / float((ti[:, 10] + ti[:, 11]).mean()),
                      "cycles_per_tile_excl_waits": (tot - float(ti[:, 1].mean()) - float(ti[:, 8].mean())) / float((ti[:, 10] + ti[:, 11]).mean()),
                      "pipeline_runs": float(ti[:, 11].mean()), "late_path_frac": float(ti[:, 12].sum() / max(1.0, float(ti[:, 10].sum())))},
       "item_block": {"mean_cycles": float(it[:, 0].mean()), "waiting_frac": float(it[:, 13].mean() / it[:, 0].mean()),
                      "input": {"n": float(it[:, 3].sum()), "cycles_each": float(it[:, 2].sum() / max(1.0, float(it[:, 3].sum())))},
                      "out": {"n": float(it[:, 5].sum()), "cycles_each": float(it[:, 4].sum() / max(1.0, float(it[:, 5].sum())))},
                      "body": {"n": float(it[:, 7].sum()), "cycles_each": float(it[:, 6].sum() / max(1.0, float(it[:, 7].sum())))},
                      "busy_frac": float((it[:, 2] + it[:, 4] + it[:, 6]).mean() / it[:, 0].mean())}}
print(json.dumps(out, indent=1))
