#!/usr/bin/env python
"""Grouped 3x3 convolutions of the RegNetY-3.2GF trunks at B=10: per-group direct kernels (csrc/conv_grouped.cpp) vs the implicit-GEMM engine."""
import os, sys, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from transfuser_amd import ops
ops.plans_load(os.path.join(ROOT, "transfuser_amd", "plans", "mi355x.txt"))
dev = "cuda"


def t(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


print("# shape (B,H,W,C)            pass     engine us   direct us   speed-up   direct TFLOP/s (algorithmic)")
for (B, H, W, C) in [(10, 64, 176, 72), (10, 64, 64, 72), (10, 32, 88, 216), (10, 32, 32, 216), (10, 16, 44, 576), (10, 16, 16, 576)]:
    g = C // 24
    x = torch.randn(B, H, W, C, device=dev)
    w = (torch.randn(C, 24, 3, 3, device=dev) * 0.1).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, H, W, C, device=dev)
    dw = torch.zeros_like(w)
    fl = 2.0 * B * H * W * C * 24 * 9
    for name, fn in (("fwd", lambda: ops.conv_fwd(x, w, None, 1, None, g)), ("dgrad", lambda: ops.conv_dgrad(dy, w, x.shape, 1, None, g)),
                     ("wgrad", lambda: ops.conv_wgrad(dy, x, dw, 1, None, g))):
        ops._GROUPED = False
        a = t(fn)
        ops._GROUPED = True
        b = t(fn)
        print("%-28s %-6s %10.1f %11.1f %9.2fx %12.1f" % ((B, H, W, C), name, a, b, a / b, fl / b / 1e6), flush=True)
