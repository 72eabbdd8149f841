#!/usr/bin/env python3
"""What bounds the PointNet's two GEMM shapes (csrc/linear.hip) at the benchmark shape: each with and without its output store (Y = NULL is a legal
call: only the column maximum is produced) and with the A operand confined to a 4 MB window (L2-resident) - timing only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egohmr_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
B, N, H = 256, 4096, 256
Np = (N + 191) // 192 * 192
M = B * Np
g = torch.Generator(device=dev).manual_seed(1)


def x2(rows, cols):
    src = torch.randn(rows, cols, device=dev, generator=g).relu_()
    dst = torch.empty(rows, cols, device=dev)
    _lib.check(L.ehm_split_pack(src.data_ptr(), dst.data_ptr(), rows, cols, cols, 1.0, None))
    return dst


def w(n, k):
    src = torch.randn(n, k, device=dev, generator=g) / k ** 0.5
    dst = torch.empty(n, k, device=dev)
    _lib.check(L.ehm_split_pack(src.data_ptr(), dst.data_ptr(), n, k, k, 256.0, None))
    return dst


cur, hb, out = x2(M, H), x2(M, H), torch.empty(M, H, device=dev)
W1, W3 = w(H, H), w(H, 2 * H)
gb = torch.randn(B, H, device=dev)
cm = torch.full((B, H), float("-inf"), device=dev)


def run(name, A0, K0, A1, K1, W, Y, colmax, relu_in0, m_rows=M):
    d = _lib.LinearDesc(A0=A0.data_ptr(), A1=A1.data_ptr() if A1 is not None else None, W=W.data_ptr(), lift_points=None, lift_W4=None, bias=None,
                        group_bias=gb.data_ptr(), Y=Y.data_ptr() if Y is not None else None, colmax=colmax.data_ptr() if colmax is not None else None,
                        M=m_rows, N=H, K0=K0, K1=K1, rows_per_group=Np, valid_rows_per_group=N, relu_in0=int(relu_in0), relu_out=int(relu_in0), w_scale=256.0)
    for _ in range(3):
        _lib.check(L.ehm_linear_split(d, None))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        _lib.check(L.ehm_linear_split(d, None))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    fl = 2.0 * m_rows * (K0 + K1) * H * 3
    print(f"{name:58s} {us:8.1f} us  {fl / us / 1e6:7.0f} TFLOP/s issued")


run("g1 (K = 256, relu in / out), output stored", cur, H, None, 0, W1, hb, None, True)
run("g1, no output store (column maximum only)", cur, H, None, 0, W1, None, cm, True)
run("g3 (K = 256 + 256), output stored + column maximum", hb, H, cur, H, W3, out, cm, False)
run("g3, no output store", hb, H, cur, H, W3, None, cm, False)
