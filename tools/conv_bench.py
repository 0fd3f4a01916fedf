#!/usr/bin/env python
"""Times conv fwd / dgrad / wgrad (HIP events) for the decoder-tail shapes through the implicit-GEMM engine and through the direct
LDS-tiled kernels.  python tools/conv_bench.py"""
import os, sys, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from transfuser_amd import ops
dev = "cuda"
def t(fn, it=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it
ops.plans_load(os.path.join(ROOT, "transfuser_amd", "plans", "mi355x.txt"))
for (B, H, W, Ci, Co) in [(10, 256, 704, 32, 32), (10, 256, 704, 32, 7), (10, 256, 704, 32, 1), (10, 64, 176, 32, 32)]:
    x = torch.randn(B, H, W, Ci, device=dev); dy = torch.randn(B, H, W, Co, device=dev)
    w = (torch.randn(Co, Ci, 3, 3, device=dev) * 0.1).contiguous(memory_format=torch.channels_last); b = torch.zeros(Co, device=dev)
    dw = torch.zeros_like(w); dx = torch.empty_like(x)
    res = {}
    for direct in (0, 1):
        ops._DIRECT = bool(direct)
        ops._DIRECT_WGRAD_MAX_COUT = 32 if direct else 4      # time the direct weight gradient for every Cout
        res[direct] = (t(lambda: ops.conv_fwd(x, w, b, 1, None, 1, relu=True)), t(lambda: ops.conv_dgrad(dy, w, x.shape, 1, None, 1, out=dx)),
                       t(lambda: ops.conv_wgrad(dy, x, dw, 1, None, 1)))
    print("%s  engine fwd/dgrad/wgrad %.0f %.0f %.0f us   direct %.0f %.0f %.0f us" % ((B, H, W, Ci, Co), *res[0], *res[1]), flush=True)
