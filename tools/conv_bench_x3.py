#!/usr/bin/env python
"""Times the direct decoder-tail / grouped 3x3 convolutions (fwd / dgrad / wgrad, HIP events) in the fp32 and f32x3 compute modes.
python tools/conv_bench_x3.py"""
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
for (B, H, W, Ci, Co, groups) in [(10, 256, 704, 32, 32, 1), (10, 256, 704, 32, 7, 1), (10, 256, 704, 32, 1, 1), (10, 16, 44, 576, 576, 24), (10, 64, 176, 72, 72, 3)]:
    x = torch.randn(B, H, W, Ci, device=dev); dy = torch.randn(B, H, W, Co, device=dev)
    w = (torch.randn(Co, Ci // groups, 3, 3, device=dev) * 0.1).contiguous(memory_format=torch.channels_last)
    dw = torch.zeros_like(w); dx = torch.empty_like(x)
    out = []
    for prec in ("fp32", "f32x3"):
        ops.set_precision(prec)
        out.append((t(lambda: ops.conv_fwd(x, w, None, 1, None, groups)), t(lambda: ops.conv_dgrad(dy, w, x.shape, 1, None, groups, out=dx)),
                    t(lambda: ops.conv_wgrad(dy, x, dw, 1, None, groups))))
    ops.set_precision("fp32")
    print("%s  fp32 fwd/dgrad/wgrad %.0f %.0f %.0f us   f32x3 %.0f %.0f %.0f us" % ((B, H, W, Ci, Co, groups), *out[0], *out[1]), flush=True)
