#!/usr/bin/env python3
"""Where FusedSampler.prepare's time goes at the benchmark shape: ResNet-50 alone, PointNet alone, both sequentially, both on two
streams, and the full prepare (encoders + projections).  GPU time by events, host time by perf_counter."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egohmr_amd import synthetic as syn  # noqa: E402
from egohmr_amd.factory import batch_to_device, build_synthetic_model  # noqa: E402

dev = torch.device("cuda:0")
model = build_synthetic_model(dev, 0)
b = batch_to_device(syn.make_batch(256, 4096, seed=100), dev)
fs = model.fused_sampler


def timed(fn, reps=20):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    t_host = (time.perf_counter() - t0) / reps * 1e3
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, t_host


img = [v for k, v in b.items() if torch.is_tensor(v) and v.dim() == 4][0]
pts = [v for k, v in b.items() if torch.is_tensor(v) and v.dim() == 3 and v.shape[-1] == 3 and v.shape[1] >= 1024][0]
backbone = fs._backbone_fn()
scene = model.scene_enc
print("image", tuple(img.shape), "scene points", tuple(pts.shape))
two_launch = model.backbone.folded(fuse_shortcut=False)
for name, fn in (("ResNet-50 trunk", lambda: backbone(img)), ("ResNet-50, shortcut convs as own launches", lambda: two_launch(img)),
                 ("ResNet-50 trunk", lambda: backbone(img)), ("ResNet-50, shortcut convs as own launches", lambda: two_launch(img)),
                 ("scene PointNet", lambda: scene(pts))):
    wall, host = timed(fn)
    print(f"{name:42s}: {wall:7.2f} ms wall, {host:6.2f} ms of it host-side issue")
for mode in (False, True, False, True, False, True):
    model.overlap_encoders = mode

    def prep():
        fs.invalidate()
        fs.prepare(b)
    wall, host = timed(prep)
    print(f"prepare overlap={mode!s:5s}: {wall:7.2f} ms wall, {host:6.2f} ms host-side issue")
