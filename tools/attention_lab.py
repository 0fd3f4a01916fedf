#!/usr/bin/env python
"""In-graph per-launch time of the fused attention kernels (csrc/attention.cpp) at the four GPT stages of the bench configuration (B = 10, T = 174, 4 heads,
C = 72 / 216 / 576 / 1512, attn_pdrop 0.1), against the algorithmic FLOPs (forward 4 T^2 hs per head, backward 2.5x).  python tools/attention_lab.py"""
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


B, T, nh = 10, 174, 4
if "--pmc" in sys.argv:       # eager launches for the counter passes of tools/pmc_attention.sh: the widest and the narrowest stage only
    seed = torch.zeros(1, dtype=torch.int32, device=dev)
    for C in (1512, 576, 72):
        qkv = torch.randn(B * T, 3 * C, device=dev) * 0.5
        dy = torch.randn(B * T, C, device=dev)
        for _ in range(6):
            y, lse = ops.attention_fwd(qkv, B, T, C, nh, drop=(seed, 7, 0.1))
            ops.attention_bwd(qkv, dy, lse, B, T, C, nh, drop=(seed, 7, 0.1))
    torch.cuda.synchronize()
    sys.exit(0)
seed = torch.zeros(1, dtype=torch.int32, device=dev)
print("# C (hs): forward us (TF/s) | backward dq + dkv us (TF/s), in-graph, best of 5 replays of %d" % REP)
tot = 0.0
for C in (72, 216, 576, 1512):
    qkv = torch.randn(B * T, 3 * C, device=dev) * 0.5
    dy = torch.randn(B * T, C, device=dev)
    y, lse = ops.attention_fwd(qkv, B, T, C, nh, drop=(seed, 7, 0.1))
    tf = graph_time(lambda: ops.attention_fwd(qkv, B, T, C, nh, drop=(seed, 7, 0.1)))
    tb2 = graph_time(lambda: ops.attention_bwd(qkv, dy, lse, B, T, C, nh, drop=(seed, 7, 0.1)))                # two dependent launches (rounds 3-5)
    tb = graph_time(lambda: ops.attention_bwd(qkv, dy, lse, B, T, C, nh, drop=(seed, 7, 0.1), y=y))            # D = dY . Y, then both halves as one grid
    fl = 4.0 * T * T * (C // nh) * nh * B
    tot += 4 * (tf + tb)
    print("C = %4d (hs %3d): forward %6.1f us (%5.1f TF/s) | backward %6.1f us (%5.1f TF/s); as two dependent launches %6.1f us" %
          (C, C // nh, tf, fl / tf / 1e6, tb, 2.5 * fl / tb / 1e6, tb2), flush=True)
print("# 4 layers per stage, forward + backward: %.2f ms per training step" % (tot / 1e3))
