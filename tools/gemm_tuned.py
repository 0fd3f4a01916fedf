#!/usr/bin/env python
"""Launches ONE GEMM shape under the shipped tuned plan (transfuser_amd/plans/mi355x.txt) - the PMC target of tools/pmc_roofline.sh.
python tools/gemm_tuned.py M N K [form nt|nn|tn] [iters] [precision fp32|f32x3|bf16|fp16]
(bf16 / fp16: the 16-bit STORED operand path of the GPT linear layers, tf_gemm16_nt_f32 - what bench.py --dtype bf16 / fp16 times)"""
import os, sys, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from transfuser_amd import ops
M, N, K = [int(v) for v in sys.argv[1:4]]
form = sys.argv[4] if len(sys.argv) > 4 else "nt"
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
ops.set_precision(sys.argv[6] if len(sys.argv) > 6 else "fp32")
ops.plans_load(os.path.join(ROOT, "transfuser_amd", "plans", "mi355x.txt"))
dev = "cuda"
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.02; b = torch.zeros(N, device=dev); out = torch.empty(M, N, device=dev)
dy = torch.randn(M, N, device=dev); dw = torch.zeros(N, K, device=dev); dx = torch.empty(M, K, device=dev)
prec = sys.argv[6] if len(sys.argv) > 6 else "fp32"
if prec in ("bf16", "fp16") and form == "nt":
    x16, _ = ops.cast16(x, want_t=False); w16, _ = ops.cast16(w, want_t=False)
    for _ in range(iters + 2):
        ops.gemm16_nt(x16, w16, out, bias=b, relu=True)
    torch.cuda.synchronize()
    sys.exit(0)
for _ in range(iters + 2):
    if form == "nt": ops.linear_fwd(x, w, b, relu=True, out=out)
    elif form == "nn": ops.linear_dgrad(dy, w, out=dx)
    else: ops.linear_wgrad(dy, x, dw, accumulate=True)
torch.cuda.synchronize()
