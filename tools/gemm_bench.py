#!/usr/bin/env python
"""Times one GEMM shape under forced tilings (HIP events); used for PMC runs.  python tools/gemm_bench.py M N K [form] [iters]"""
import os, sys, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from transfuser_amd import ops
M, N, K = [int(v) for v in sys.argv[1:4]]
form = sys.argv[4] if len(sys.argv) > 4 else "nt"
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
dev = "cuda"
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.02; b = torch.zeros(N, device=dev); out = torch.empty(M, N, device=dev)
dy = torch.randn(M, N, device=dev); dw = torch.zeros(N, K, device=dev); dx = torch.empty(M, K, device=dev)
def run():
    if form == "nt": ops.linear_fwd(x, w, b, relu=True, out=out)
    elif form == "nn": ops.linear_dgrad(dy, w, out=dx)
    else: ops.linear_wgrad(dy, x, dw, accumulate=True)
for plan in [(128, 128, 16), (128, 128, 32), (128, 96, 16), (128, 64, 16), (64, 128, 16), (64, 64, 16), (64, 64, 32)]:
    ops.force_plan(*plan, 1)
    for _ in range(2): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print("%s %dx%dx%d plan %s: %.1f us  %.1f TFLOP/s" % (form, M, N, K, plan, us, 2.0 * M * N * K / us / 1e6), flush=True)
ops.force_plan(0)
