#!/usr/bin/env python3
"""The vendor libraries on the two step-invariant encoders at the benchmark shape (B = 256 images 224 x 224, 256 x 4096 scene points): eager PyTorch =
MIOpen convolutions / hipBLASLt GEMMs, in float32 (the precision class the repository's split-f16 kernels deliver) and under float16 autocast (NOT a parity
path: a cost reference), against `ResNet50Features.folded()` / `ResnetPointnet.forward` of this package.  (docs/EXPERIMENTS.md 3.3 / 3.4)

    MIOPEN_FIND_MODE=FAST python tools/encoder_yardstick.py
"""
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egohmr_amd.encoders import ResNet50Features, ResnetPointnet  # noqa: E402
from _eager import resnet50_eager  # noqa: E402

dev = torch.device("cuda:0")
B, N = 256, 4096
torch.manual_seed(0)


def timed(fn, secs=2.0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    spans, t_end = [], time.time() + secs
    while time.time() < t_end:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        spans.append(e0.elapsed_time(e1))
    return sum(spans[len(spans) // 2:]) / len(spans[len(spans) // 2:])


def eager_pointnet(net, pts):                       # respointnet.py:33-59 as plain torch ops on the module's parameters
    def block(b, x):
        h = b.fc_0(F.relu(x))
        return b.shortcut(x) + b.fc_1(F.relu(h))
    x = block(net.block_0, net.fc_pos_0(pts))
    for b in (net.block_1, net.block_2, net.block_3):
        x = block(b, torch.cat([x, x.max(dim=1, keepdim=True)[0].expand(x.size())], dim=2))
    return net.fc_c(F.relu(x.max(dim=1)[0]))


with torch.no_grad():
    rn = ResNet50Features().to(dev).eval()
    img = torch.randn(B, 3, 224, 224, device=dev)
    img_cl = img.contiguous(memory_format=torch.channels_last)
    rn_cl = ResNet50Features().to(dev).eval().to(memory_format=torch.channels_last)
    rn_cl.load_state_dict(rn.state_dict())
    ours = rn.folded()
    ref = resnet50_eager(rn, img)
    err = float((ours(img) - ref).abs().max() / ref.abs().max())
    rows = [("resnet50 B256: this package (split-f16 X2 trunk, f32-grade)", timed(lambda: ours(img))),
            ("resnet50 B256: eager float32 NCHW (MIOpen)", timed(lambda: resnet50_eager(rn, img))),
            ("resnet50 B256: eager float32 channels_last (MIOpen)", timed(lambda: resnet50_eager(rn_cl, img_cl)))]
    with torch.autocast("cuda", dtype=torch.float16):
        rows.append(("resnet50 B256: eager float16 autocast channels_last (MIOpen; not f32-grade)", timed(lambda: resnet50_eager(rn_cl, img_cl))))
    for name, ms in rows:
        print(json.dumps({"what": name, "ms": round(ms, 3), "ours_vs_eager_f32_rel_err": err}), flush=True)
    pn = ResnetPointnet().to(dev).eval()
    pts = torch.randn(B, N, 3, device=dev)
    ref = eager_pointnet(pn, pts)
    err = float((pn(pts) - ref).abs().max() / ref.abs().max())
    rows = [("pointnet 256x4096: this package (split-f16, f32-grade)", timed(lambda: pn(pts))),
            ("pointnet 256x4096: eager float32 (hipBLASLt)", timed(lambda: eager_pointnet(pn, pts)))]
    with torch.autocast("cuda", dtype=torch.float16):
        rows.append(("pointnet 256x4096: eager float16 autocast (hipBLASLt; not f32-grade)", timed(lambda: eager_pointnet(pn, pts))))
    for name, ms in rows:
        print(json.dumps({"what": name, "ms": round(ms, 3), "ours_vs_eager_f32_rel_err": err}), flush=True)
