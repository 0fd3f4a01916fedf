#!/usr/bin/env python
"""Where does a small GEMM launch spend its time inside a replayed hipGraph?  Back-to-back launches of the engine on the trunk shapes with the
contraction length K swept down to one k-tile (the K -> 0 intercept is the per-launch fixed cost: dispatch gap, kernarg / index prologue,
first operand round trip, epilogue, drain) and the output size swept at fixed K (does the fixed cost scale with the bytes stored?), beside the
floor of a trivial kernel.  python tools/launch_lab.py"""
import os, sys, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from transfuser_amd import ops
ops.plans_load(os.path.join(ROOT, "transfuser_amd", "plans", "mi355x.txt"))
dev = "cuda"
REP = 40


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


a = torch.randn(1 << 20, device=dev); b = torch.empty_like(a)
print("floor: relu_mask 4 KB %.2f us, 4 MB %.2f us" % (graph_time(lambda: ops.relu_mask(a[:1024], a[:1024], b[:1024])), graph_time(lambda: ops.relu_mask(a, a, b))), flush=True)
for (M, N) in [(2560, 576), (7040, 576), (10240, 216), (28160, 216), (640, 576), (1740, 1512)]:
    line = "fwd nt %6d x %4d :" % (M, N)
    for K in (32, 64, 128, 256, 576, 1152):
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.02; out = torch.empty(M, N, device=dev)
        ops.force_plan(64, 64, 32, 1)
        t = graph_time(lambda: ops.gemm(x, w, out, M, N, K, K, K, N))
        ops.force_plan(0)
        line += "  K=%d %.1f" % (K, t)
    print(line, flush=True)
# the same launches on TWO streams inside one graph (image / LiDAR trunk pairing): is the pair's time the sum or the max?
M1, M2, N, K = 7040, 2560, 576, 576
x1 = torch.randn(M1, K, device=dev); x2 = torch.randn(M2, K, device=dev); w1 = torch.randn(N, K, device=dev) * 0.02; w2 = torch.randn(N, K, device=dev) * 0.02
o1 = torch.empty(M1, N, device=dev); o2 = torch.empty(M2, N, device=dev)
f1 = lambda: ops.gemm(x1, w1, o1, M1, N, K, K, K, N)
f2 = lambda: ops.gemm(x2, w2, o2, M2, N, K, K, K, N)
t1, t2 = graph_time(f1), graph_time(f2)
for _ in range(3): f1(); f2()
torch.cuda.synchronize()
s = torch.cuda.Stream(); s2 = torch.cuda.Stream()
with torch.cuda.stream(s):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        ev = torch.cuda.Event(); ev.record(s); s2.wait_event(ev)
        for _ in range(REP): f1()
        with torch.cuda.stream(s2):
            for _ in range(REP): f2()
            ev2 = torch.cuda.Event(); ev2.record(s2)
        s.wait_event(ev2)
    g.replay(); torch.cuda.synchronize()
    best = 1e30
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s); g.replay(); e1.record(s); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / REP)
print("two branches of one graph: 7040-row GEMM alone %.1f us, 2560-row alone %.1f us, one of each per pair on two streams %.1f us (sum %.1f, max %.1f)" %
      (t1, t2, best, t1 + t2, max(t1, t2)), flush=True)
# one launch over the concatenated rows (the bound of a merged image + LiDAR launch)
xc = torch.randn(M1 + M2, K, device=dev); oc = torch.empty(M1 + M2, N, device=dev)
print("one launch over %d rows: %.1f us" % (M1 + M2, graph_time(lambda: ops.gemm(xc, w1, oc, M1 + M2, N, K, K, K, N))), flush=True)
