#!/usr/bin/env python
"""In-graph per-launch times of the trunk 1x1-convolution GEMMs under the shipped plans: forward, input gradient, weight gradient, the
(weight gradient, input gradient) pair as two launches and as ONE grid (ops.gemm_pair, csrc/gemm_pair.cpp).  Run once per experiment switch
(TF_GEMM_PF2=1: prefetch distance 2 in the 64 x 64 tiles; TF_GEMM_STAGGER=1: distinct issue priorities for co-resident workgroups).
python tools/pair_lab.py"""
import os, sys, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from transfuser_amd import ops
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
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); g.replay(); e1.record(s); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / REP)
    return best


print("# switches: TF_GEMM_PF2=%s TF_GEMM_STAGGER=%s" % (os.environ.get("TF_GEMM_PF2", "0"), os.environ.get("TF_GEMM_STAGGER", "0")))
print("# shape: fwd nt | dgrad nn | wgrad tn | wgrad + dgrad as two launches | as one grid (us, in-graph, best of 5 replays of %d)" % REP)
for (M, N, K) in [(7040, 576, 576), (2560, 576, 576), (28160, 216, 216), (10240, 216, 216), (112640, 72, 72), (1740, 576, 2304), (1740, 216, 864)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.02; out = torch.empty(M, N, device=dev)
    dy = torch.randn(M, N, device=dev); dw = torch.zeros(N, K, device=dev); dx = torch.empty(M, K, device=dev)

    def seq():
        ops.linear_wgrad(dy, x, dw); ops.linear_dgrad(dy, w, out=dx)

    def pair():
        with ops.gemm_pair(dy):
            ops.linear_wgrad(dy, x, dw); ops.linear_dgrad(dy, w, out=dx)
    p0 = ops.gemm_pair_count()
    pair()
    joint = ops.gemm_pair_count() > p0
    cs = ops.ColStat(M, N, dev)
    t = [graph_time(lambda: ops.linear_fwd(x, w, out=out)), graph_time(lambda: ops.linear_dgrad(dy, w, out=dx)), graph_time(lambda: ops.linear_wgrad(dy, x, dw)),
         graph_time(seq), graph_time(pair), graph_time(lambda: ops.gemm(x, w, out, M, N, K, K, K, N, colstat=cs))]
    fl = 2.0 * M * N * K
    print("%-20s fwd %6.1f (%5.1f TF/s) | fwd + BN statistics %6.1f | dgrad %6.1f | wgrad %6.1f | two launches %6.1f | one grid %6.1f%s" % (
        (M, N, K), t[0], fl / t[0] / 1e6, t[5], t[1], t[2], t[3], t[4], "" if joint else "  (no joint kernel for these plans)"), flush=True)
