#!/usr/bin/env python
"""Stream-K vs data-parallel LDS-DMA plans on the GPT shapes, in-graph per-launch time (a replayed hipGraph of back-to-back launches).
python tools/sk_lab.py [precision]"""
import os, sys, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from transfuser_amd import ops
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
ops.set_precision(prec)
ops.plans_load(os.path.join(ROOT, "transfuser_amd", "plans", "mi355x.txt"))
dev = "cuda"
REP = 10
SK = 2000000


def graph_time(fn, rep=REP):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(rep): fn()
        g.replay(); torch.cuda.synchronize()
        best = 1e30
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); g.replay(); e1.record(s); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / rep)
    return best


shapes = [(1740, 6048, 1512), (1740, 1512, 6048), (1740, 4536, 1512), (1740, 1512, 1512), (1740, 2304, 576), (1740, 576, 2304), (7040, 576, 576), (28160, 216, 216), (7040, 1512, 576)]
for (M, N, K) in shapes:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.02; b = torch.zeros(N, device=dev); out = torch.empty(M, N, device=dev)
    dy = torch.randn(M, N, device=dev); dw = torch.zeros(N, K, device=dev); dx = torch.empty(M, K, device=dev)
    forms = {"fwd nt": lambda: ops.linear_fwd(x, w, b, relu=True, out=out), "dgrad nn": lambda: ops.linear_dgrad(dy, w, out=dx),
             "wgrad tn": lambda: ops.linear_wgrad(dy, x, dw, accumulate=True)}
    fl = 2.0 * M * N * K
    for name, fn in forms.items():
        t = graph_time(fn)
        line = "%s %-9s %6dx%5dx%5d: tuned %6.1f us %5.0f TF/s |" % (prec, name, M, N, K, t, fl / t / 1e6)
        for kind in (1, 3, 4, 5, 2, 8, 7):
            ops.force_dma(kind, 1); tp = graph_time(fn)
            ops.force_dma(kind, SK); ts = graph_time(fn)
            line += " d%d %.0f/sk %.0f" % (kind, tp, ts)
        ops.force_plan(0)
        print(line, flush=True)
