#!/usr/bin/env python
"""In-graph per-launch time of the grouped 3x3 kernels (csrc/conv_grouped.cpp) on the trunk shapes, against their fp32-MFMA bound
(2 * pixels * C * 24 * 9 flops at 75 % column use for fwd / dgrad: 32 of 24 columns multiplied).  python tools/grouped_lab.py"""
import os, sys, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from transfuser_amd import ops
dev = "cuda"
REP = 20


def graph_time(fn, rep=REP):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(rep): fn()
        g.replay(); torch.cuda.synchronize()
        best = 1e30
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); g.replay(); e1.record(s); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / rep)
    return best


print("# shape (B,H,W,C)  pass: in-graph us | MFMA-issue bound us at 2.05 GHz (32x32x2 MFMAs actually issued) | algorithmic TFLOP/s")
for (B, H, W, C) in [(10, 16, 44, 576), (10, 16, 16, 576), (10, 32, 88, 216), (10, 32, 32, 216), (10, 64, 176, 72), (10, 64, 64, 72), (10, 8, 22, 1512), (10, 8, 8, 1512)]:
    g = C // 24
    x = torch.randn(B, H, W, C, device=dev)
    w = (torch.randn(C, 24, 3, 3, device=dev) * 0.1).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, H, W, C, device=dev)
    dw = torch.zeros_like(w)
    fl = 2.0 * B * H * W * C * 24 * 9
    tw = 16 if (-(-H // 8) * 8) * (-(-W // 16) * 16) < (-(-H // 4) * 4) * (-(-W // 32) * 32) else 32
    th = 8 if tw == 16 else 4
    ntiles = B * -(-H // th) * -(-W // tw) * g
    bound = ntiles * 4 * 108 * 64 / 1024 / 2050.0           # tiles x 4 waves x 108 MFMAs x 64 clk / 1024 SIMDs / MHz
    for name, fn, mf in (("fwd", lambda: ops.conv_fwd(x, w, None, 1, None, g), 1.0), ("fwd+stat", lambda: ops.conv_fwd(x, w, None, 1, None, g, colstat=True), 1.0),
                         ("dgrad", lambda: ops.conv_dgrad(dy, w, x.shape, 1, None, g), 1.0), ("wgrad", lambda: ops.conv_wgrad(dy, x, dw, 1, None, g), 576.0 / 432.0)):
        t = graph_time(fn)
        print("%-22s %-8s %7.1f us | %6.1f us | %6.1f TF/s" % ((B, H, W, C), name, t, bound * mf, fl / t / 1e6), flush=True)
