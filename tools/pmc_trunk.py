#!/usr/bin/env python
"""PMC targets of tools/pmc_trunk.sh (round-4 verdict, next #2a): the contractions with the most step time that had no counter evidence - the
RegNetY 1x1-convolution GEMMs of stage 3 / stage 2 (image and LiDAR trunk row counts; forward nt, input gradient nn, weight gradient tn) under the
shipped plans, and the three grouped 3x3 kernels at (10, 16, 44, 576).  Cases run back to back, ITERS launches each, separated by a marker
launch (torch cos_ on a small tensor: a kernel the path itself never launches) that tools/pmc_trunk_summary.py uses to cut the dispatch list.
python tools/pmc_trunk.py [list]   (list: print the case table as JSON and exit)"""
import json, os, sys, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
ITERS = 8
GEMMS = [(7040, 576, 576), (2560, 576, 576), (28160, 216, 216), (10240, 216, 216)]
GROUPED = [(10, 16, 44, 576)]


def cases():
    out = []
    for (M, N, K) in GEMMS:
        for form in ("nt", "nn", "tn"):
            byt = (M * K + N * K + M * N) * 4
            out.append(dict(name="gemm %s (%d,%d,%d)" % (form, M, N, K), flops=2.0 * M * N * K, bytes=byt, match="gemm"))
        # the layer's weight + input gradient as ONE grid (csrc/gemm_pair.cpp): dy read by both, x and W read once, dx and dW written
        out.append(dict(name="gemm pair (%d,%d,%d)" % (M, N, K), flops=4.0 * M * N * K, bytes=(2 * M * N + 2 * M * K + 2 * N * K) * 4, match="gemm"))
    for (B, H, W, C) in GROUPED:
        E = B * H * W * C
        for p in ("fwd", "dgrad", "wgrad"):
            out.append(dict(name="grouped3x3 %s %s" % (p, (B, H, W, C)), flops=2.0 * E * 24 * 9, bytes=(2 * E + C * 24 * 9) * 4, match="conv3x3_grouped"))
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "list":
        print(json.dumps(cases()))
        sys.exit(0)
    from transfuser_amd import ops
    ops.set_precision("fp32")
    ops.plans_load(os.path.join(ROOT, "transfuser_amd", "plans", "mi355x.txt"))
    dev = "cuda"
    fns = []
    for (M, N, K) in GEMMS:
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; y = torch.empty(M, N, device=dev)
        dy = torch.randn(M, N, device=dev); dw = torch.zeros(N, K, device=dev); dx = torch.empty(M, K, device=dev)
        fns.append(lambda x=x, w=w, y=y: ops.linear_fwd(x, w, out=y))
        fns.append(lambda dy=dy, w=w, dx=dx: ops.linear_dgrad(dy, w, out=dx))
        fns.append(lambda dy=dy, x=x, dw=dw: ops.linear_wgrad(dy, x, dw, accumulate=True))

        def pair(dy=dy, x=x, dw=dw, w=w, dx=dx):
            with ops.gemm_pair(dy):
                ops.linear_wgrad(dy, x, dw, accumulate=True)
                ops.linear_dgrad(dy, w, out=dx)
        fns.append(pair)
    for (B, H, W, C) in GROUPED:
        g = C // 24
        x = torch.randn(B, H, W, C, device=dev); dy = torch.randn(B, H, W, C, device=dev)
        w = (torch.randn(C, 24, 3, 3, device=dev) * 0.1).contiguous(memory_format=torch.channels_last); dw = torch.zeros_like(w)
        fns.append(lambda x=x, w=w, g=g: ops.conv_fwd(x, w, None, 1, None, g))
        fns.append(lambda dy=dy, w=w, x=x, g=g: ops.conv_dgrad(dy, w, x.shape, 1, None, g))
        fns.append(lambda dy=dy, x=x, dw=dw, g=g: ops.conv_wgrad(dy, x, dw, 1, None, g))
    for fn in fns:      # first-use work (workspaces, plan lookups) outside the marked segments
        fn()
    torch.cuda.synchronize()
    marks = [torch.zeros(4096, device=dev) for i in range(len(fns) + 1)]
    torch.cuda.synchronize()
    for i, fn in enumerate(fns):
        marks[i].cos_()
        for _ in range(ITERS):
            fn()
    marks[-1].cos_()
    torch.cuda.synchronize()
