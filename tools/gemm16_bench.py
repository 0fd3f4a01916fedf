#!/usr/bin/env python
"""Packed 16-bit NT GEMM (tf_gemm16_nt_f32) on the GPT shapes, every LDS-DMA tile configuration: us / TFLOP/s per (shape, kind).
python tools/gemm16_bench.py [bf16|fp16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transfuser_amd import ops

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
ops.set_precision(mode)
dev = "cuda"
# (m, n, k, accumulate): forward / input-gradient products of the GPT-4 / GPT-3 Blocks at B = 10 (T = 174), and their weight gradients (k = 1744 = rows padded to 8, accumulate)
shapes = [(1740, 6048, 1512, 0), (1740, 1512, 6048, 0), (1740, 4536, 1512, 0), (1740, 1512, 4536, 0), (1740, 1512, 1512, 0),
          (1512, 6048, 1744, 1), (6048, 1512, 1744, 1), (4536, 1512, 1744, 1), (1512, 1512, 1744, 1),
          (1740, 2304, 576, 0), (1740, 576, 2304, 0), (1740, 1728, 576, 0), (1740, 576, 1728, 0), (1740, 576, 576, 0),
          (576, 2304, 1744, 1), (2304, 576, 1744, 1), (1728, 576, 1744, 1), (576, 576, 1744, 1),
          (1740, 864, 216, 0), (1740, 216, 864, 0), (7040, 576, 576, 0), (4096, 4096, 4096, 0)]
if len(sys.argv) > 2:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]]
print("# %s; columns: kind 0 = library heuristic, 1..8 = pinned LDS-DMA configuration" % mode)
for (m, n, k, acc) in shapes:
    a = torch.randn(m, k, device=dev); b = torch.randn(n, k, device=dev)
    a16, _ = ops.cast16(a, want_t=False); b16, _ = ops.cast16(b, want_t=False)
    out = torch.empty(m, n, device=dev)
    row = []
    for kind in range(0, 9):
        for _ in range(3):
            ops.gemm16_nt(a16, b16, out, kind=kind, accumulate=bool(acc))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm16_nt(a16, b16, out, kind=kind, accumulate=bool(acc))
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        row.append("%d:%6.1fus %5.0fTF" % (kind, us, 2.0 * m * n * k / us / 1e6))
    print("(%5d,%5d,%5d)%s " % (m, n, k, "+" if acc else " ") + "  ".join(row))
