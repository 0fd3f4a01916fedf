#!/usr/bin/env python
"""In-graph per-launch time of the trunk 1x1-convolution GEMM shapes: the tuned plan and every forced tiling, with / without the fused BatchNorm
statistics.  A captured hipGraph of REP back-to-back launches is replayed (no host launch gaps: the number a training step pays).
python tools/shape_lab.py [precision] [shape,shape,...]     shape = MxNxK"""
import os, sys, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from transfuser_amd import ops
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
shapes = [tuple(int(v) for v in s.split("x")) for s in (sys.argv[2].split(",") if len(sys.argv) > 2 else
          ["7040x576x576", "2560x576x576", "28160x216x216", "10240x216x216", "112640x72x72", "1760x1512x1512"])]
ops.set_precision(prec)
ops.plans_load(os.path.join(ROOT, "transfuser_amd", "plans", "mi355x.txt"))
dev = "cuda"
REP = 20


def graph_time(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(REP): fn()
        g.replay(); torch.cuda.synchronize()
        best = 1e30
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); g.replay(); e1.record(s); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / REP)
    return best


for (M, N, K) in shapes:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.02; out = torch.empty(M, N, device=dev)
    dy = torch.randn(M, N, device=dev); dw = torch.zeros(N, K, device=dev); dx = torch.empty(M, K, device=dev)
    cs = ops.ColStat(M, N, dev)
    forms = {
        "fwd nt +stat": lambda: ops.gemm(x, w, out, M, N, K, K, K, N, colstat=cs),
        "fwd nt": lambda: ops.gemm(x, w, out, M, N, K, K, K, N),
        "dgrad nn": lambda: ops.linear_dgrad(dy, w, out=dx),
        "wgrad tn": lambda: ops.linear_wgrad(dy, x, dw, accumulate=True),
    }
    fl = 2.0 * M * N * K
    for name, fn in forms.items():
        t = graph_time(fn)
        line = "%s %-13s %6dx%4dx%5d: tuned %6.1f us %5.0f TF/s |" % (prec, name, M, N, K, t, fl / t / 1e6)
        sk = 6 if name.startswith("wgrad") else 1
        for plan in [(64, 64, 16), (64, 64, 32), (128, 64, 16), (128, 64, 32), (128, 128, 32)]:
            ops.force_plan(*plan, sk)
            line += " r%dx%dx%d %.1f" % (plan + (graph_time(fn),))
        ops.force_plan(0)
        for kind in range(1, 9):
            ops.force_dma(kind, sk)
            line += " d%d %.1f" % (kind, graph_time(fn))
        ops.force_plan(0)
        print(line, flush=True)
