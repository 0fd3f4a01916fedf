#!/usr/bin/env python
"""fp32 MFMA vs bf16x3-split ("f32x3") vs bf16 on the step's big plain-GEMM shapes: HIP-event time per LDS-DMA configuration and the error of
each precision against a float64 product (computed on the host).  python tools/x3_bench.py [iters]"""
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from transfuser_amd import ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = "cuda"
SHAPES = [("nt", 1740, 6048, 1512), ("nn", 1740, 1512, 6048), ("tn", 6048, 1512, 1740), ("nt", 1740, 4536, 1512), ("nt", 7040, 576, 576),
          ("tn", 576, 576, 7040), ("nt", 2560, 576, 576), ("nt", 28160, 216, 216), ("nt", 4096, 4096, 4096)]
KINDS = {1: "128x128x16", 2: "64x64x16", 3: "128x64x16", 4: "64x128x16", 5: "128x128x32", 6: "64x64 w1", 7: "64x128 w2", 8: "128x64 w2"}


def make(form, m, n, k):
    g = torch.Generator().manual_seed(m * 7 + n * 3 + k)
    if form == "nt":     # y = x w^T
        x, w = torch.randn(m, k, generator=g), torch.randn(n, k, generator=g) * 0.05
        out = torch.empty(m, n, device=dev)
        xd, wd = x.to(dev), w.to(dev)
        return (lambda: ops.linear_fwd(xd, wd, None, out=out)), (lambda: x.double() @ w.double().t())
    if form == "nn":     # dx = dy w
        dy, w = torch.randn(m, k, generator=g), torch.randn(k, n, generator=g) * 0.05
        out = torch.empty(m, n, device=dev)
        dyd, wd = dy.to(dev), w.to(dev)
        return (lambda: ops.linear_dgrad(dyd, wd, out=out)), (lambda: dy.double() @ w.double())
    dy, x = torch.randn(k, m, generator=g), torch.randn(k, n, generator=g)     # dw = dy^T x
    dw = torch.zeros(m, n, device=dev)
    dyd, xd = dy.to(dev), x.to(dev)
    return (lambda: ops.linear_wgrad(dyd, xd, dw, accumulate=False)), (lambda: dy.double().t() @ x.double())


def timed(run):
    for _ in range(2):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for form, m, n, k in SHAPES:
    run, ref64 = make(form, m, n, k)
    ref = ref64() if m * n * k <= 20e9 else None
    for prec in ("fp32", "f32x3", "bf16"):
        ops.set_precision(prec)
        best = None
        for kind in KINDS:
            ops.force_dma(kind, 1)
            us = timed(run)
            if best is None or us < best[0]:
                best = (us, kind)
            print("  %s %5dx%5dx%5d %-5s dma %-10s %8.1f us %7.1f TFLOP/s" % (form, m, n, k, prec, KINDS[kind], us, 2.0 * m * n * k / us / 1e6), flush=True)
        ops.force_dma(best[1], 1)
        got = run()
        err = ""
        if ref is not None:
            e = ((got.double().cpu() - ref).abs().max() / ref.abs().max()).item()
            err = "max rel err vs fp64 %.2e" % e
        print("%s %dx%dx%d %-5s BEST dma %-10s %8.1f us %7.1f TFLOP/s  %s" % (form, m, n, k, prec, KINDS[best[1]], best[0], 2.0 * m * n * k / best[0] / 1e6, err), flush=True)
ops.force_plan(0)
ops.set_precision("fp32")
