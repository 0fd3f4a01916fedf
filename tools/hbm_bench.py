#!/usr/bin/env python
"""Achieved HBM GB/s of the bandwidth-bound kernels north_star names (LayerNorm, softmax, BatchNorm, the H1 LiDAR histogram scatter, the
H2 pillar index scan, AdamW) at the bench workload's shapes, HIP-event timed over repeated launches, against the algorithmic bytes of
SURVEY.md section 8(d).

  python tools/hbm_bench.py > profiles/r03_hbm_kernels.txt
  python tools/hbm_bench.py --pmc --iters 8 --meta cases.json     (under rocprofv3 --pmc: tools/pmc_hbm.sh)

--pmc: a marker kernel (tf_sigmoid_f32 on one element, used by no case) is launched before and after every case so tools/pmc_hbm_summary.py can
cut the dispatch list of the counter CSV into cases (and drop the set-up kernels between two cases); --meta writes {case name, algorithmic bytes, calls} in launch order."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from transfuser_amd import ops  # noqa: E402
from transfuser_amd.data import synthetic_cloud  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pmc", action="store_true")
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--meta", default=None)
args = ap.parse_args()
dev = "cuda"
PEAK, ACHIEVABLE = 8.0e12, 6.3e12
WARM = 5
meta = []
_marker = torch.zeros(1, device=dev)


def timeit(fn, iters):
    if args.pmc:
        ops.sigmoid(_marker)
    for _ in range(WARM):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    if args.pmc:
        ops.sigmoid(_marker)      # end-of-case marker: the set-up kernels of the next case fall between two markers and are dropped
    return e0.elapsed_time(e1) * 1e-3 / iters


def row(name, bytes_, fn, iters=None):
    iters = iters or args.iters
    sec = timeit(fn, iters)
    meta.append(dict(name=name, algorithmic_bytes=bytes_, calls=WARM + iters, event_us=sec * 1e6))
    print("%-74s %9.2f MB %9.1f us %8.2f TB/s  %5.1f %% of 8.0 TB/s  %5.1f %% of 6.3 achievable" %
          (name, bytes_ / 1e6, sec * 1e6, bytes_ / sec / 1e12, 100 * bytes_ / sec / PEAK, 100 * bytes_ / sec / ACHIEVABLE), flush=True)


print("# bandwidth-bound kernels at the B=10, 256x704 workload: algorithmic bytes (SURVEY.md 8d) / HIP-event time per call, back-to-back calls")
print("# (working sets below ~200 MB stay in the 256 MB Infinity Cache between launches, so small kernels measure on-die bandwidth + launch latency)")
for C in (1512, 576, 216, 72):
    M = 1740
    x, g, b = torch.randn(M, C, device=dev), torch.ones(C, device=dev), torch.zeros(C, device=dev)
    row("layernorm_fwd  [1740 x %d]  (2 M C 4 B)" % C, 2 * M * C * 4, lambda: ops.layernorm_fwd(x, g, b))
    y, m, r = ops.layernorm_fwd(x, g, b)
    dg, db, dy = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.randn(M, C, device=dev)
    row("layernorm_bwd  [1740 x %d]  (3 M C 4 B)" % C, 3 * M * C * 4, lambda: ops.layernorm_bwd(dy, x, g, m, r, dg, db))
T, Tp, rows = 174, 176, 10 * 4 * 174
att = torch.randn(10 * 4, T, Tp, device=dev)
row("softmax_fwd    [B*4*174 x 174]  (2 B 4 174^2 4 B)", 2 * rows * T * 4, lambda: ops.softmax_fwd_(att, rows, T, Tp))
p, dp = torch.softmax(torch.randn(10 * 4, T, Tp, device=dev), -1), torch.randn(10 * 4, T, Tp, device=dev)
row("softmax_bwd    [B*4*174 x 174]  (3 B 4 174^2 4 B)", 3 * rows * T * 4, lambda: ops.softmax_bwd_(p, dp, rows, T, Tp))
for shape in ((10, 128, 352, 32), (10, 64, 176, 72), (10, 16, 44, 576)):
    x = torch.randn(*shape, device=dev)
    C = shape[-1]
    g, b, rm, rv = torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.ones(C, device=dev)
    E = x.numel()
    row("batchnorm fwd (stats + apply + ReLU) %s  (3 E 4 B)" % (shape,), 3 * E * 4, lambda: ops.bn_fwd(x, g, b, rm, rv, None, True))
    y, sm, si = ops.bn_fwd(x, g, b, rm, rv, None, True)
    dz, dg, db = torch.randn_like(x), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    row("batchnorm bwd (reduce + apply)       %s  (5 E 4 B)" % (shape,), 5 * E * 4, lambda: ops.bn_bwd(dz, y, x, g, sm, si, dg, db))
pts = torch.from_numpy(synthetic_cloud(10, 32768, 0)).to(dev)
row("H1 lidar_hist  10 x 32768 points  (N 16 B read + 512 KB written / sample)", 10 * (32768 * 16 + 2 * 256 * 256 * 4), lambda: ops.lidar_hist(pts))
raw = torch.zeros(10, 40000, 4, device=dev); raw[:, :32768] = pts
num = torch.full((10,), 32768, dtype=torch.int32, device=dev)
row("H2 pillar_index 10 x 40000 points (16 B / point + 4 B / grid cell read)", 10 * (32768 * 16 + 257 * 257 * 4),
    lambda: ops.pillar_index(raw, num, -16, 16, -32, 0, 8), iters=min(10, args.iters))
# round 6: the same call priced WITH the outputs it has to write (compacted cloud 16 B, inverse index 4 B, nine features 36 B per kept point, 4 B per pillar),
# its static-shape (hipGraph) mode - no host read, capacity-sized outputs - and the seven-launch form of rounds 3-5 beside it
_ix = ops.pillar_index(raw, num, -16, 16, -32, 0, 8)
_h2 = 10 * 32768 * 16 + _ix["N"] * 56 + _ix["P"] * 4
row("H2 pillar_index, %d kept points / %d pillars (16 B / point read + 56 B / kept point + 4 B / pillar written)" % (_ix["N"], _ix["P"]), _h2,
    lambda: ops.pillar_index(raw, num, -16, 16, -32, 0, 8), iters=min(10, args.iters))
row("H2 pillar_index, static shapes (same bytes + the zero tail: 56 B / dropped point, 4 B / unused pillar slot)", 10 * 32768 * 16 + 400000 * 56 + 400000 * 4,
    lambda: ops.pillar_index(raw, num, -16, 16, -32, 0, 8, static=True), iters=min(10, args.iters))
row("H2 pillar_index, seven-launch form of rounds 3-5 (TF_PILLAR_V2=0)", _h2,
    lambda: ops._pillar_index_v1(raw, num, -16, 16, -32, 0, 8), iters=min(10, args.iters))
n = 168018327
# ONE allocation cut into 4 arenas, like train.ParamArena / FlatAdamW (4 separate torch.randn(n) tensors were what r02 timed)
buf = torch.randn(4, (n + 63) // 64 * 64, device=dev) * 0.01
p_, g_, m_, v_ = buf[0, :n], buf[1, :n], buf[2, :n], buf[3, :n]
v_.abs_()
st = torch.tensor([0.0, 1e-4], device=dev)
row("adamw_kernel   168.0 M parameters  (28 B / parameter)", 28 * n, lambda: ops.adamw_(p_, g_, m_, v_, st), iters=min(10, args.iters))
x = torch.randn(10 * 174, 1512, device=dev)
seed = torch.zeros(1, dtype=torch.int32, device=dev)
row("dropout_kernel [1740 x 1512]  (2 M C 4 B)", 2 * x.numel() * 4, lambda: ops.dropout(x, seed, 3, 0.1))
# round 3: the decoders' thin-output tail (csrc/conv_thin.cpp), the 16-bit operand casts (csrc/cast16.cpp) and the ResNet max pool
xt = torch.randn(10, 256, 704, 32, device=dev)
for co in (7, 1):
    wt = (torch.randn(co, 32, 3, 3, device=dev) * 0.1).contiguous(memory_format=torch.channels_last)
    bt = torch.zeros(co, device=dev)
    E, Eo = xt.numel(), 10 * 256 * 704 * co
    row("conv3x3 thin fwd   32 -> %d at 10 x 256 x 704  (read x, write y)" % co, (E + Eo) * 4, lambda: ops.conv_fwd(xt, wt, bt, 1, None, 1), iters=min(10, args.iters))
    dyt = torch.randn(10, 256, 704, co, device=dev)
    dxt = torch.empty_like(xt)
    row("conv3x3 thin dgrad 32 -> %d (+ ReLU mask)  (read dy, mask, write dx)" % co, (2 * E + Eo) * 4, lambda: ops.conv_dgrad(dyt, wt, xt.shape, 1, None, 1, out=dxt, mask=xt),
        iters=min(10, args.iters))
    dwt, dbt = torch.zeros_like(wt), torch.zeros(co, device=dev)
    row("conv3x3 thin wgrad 32 -> %d (+ bias grad)  (read x, dy)" % co, (E + Eo) * 4, lambda: ops.conv_wgrad(dyt, xt, dwt, 1, None, 1, dbias=dbt), iters=min(10, args.iters))
del xt, dxt
ops.set_precision("bf16")
for (M, C) in ((1740, 6048), (1740, 1512)):
    xc = torch.randn(M, C, device=dev)
    row("cast16 (row-major + transposed bf16 copies) [%d x %d]  (4 B read + 2 x 2 B written)" % (M, C), M * C * 8, lambda: ops.cast16(xc))
# round 5 (second session): the producers that write the 16-bit operand copies themselves (norm.cpp layernorm_fwd16, reduce.cpp tile16_kernel)
for (M, C) in ((1740, 1512), (1740, 576)):
    xl, gl, bl_ = torch.randn(M, C, device=dev), torch.ones(C, device=dev), torch.zeros(C, device=dev)
    row("layernorm -> bf16 copies (row-major + transposed) [%d x %d]  (4 B read + 2 x 2 B written)" % (M, C), M * C * 8, lambda: ops.layernorm_fwd16(xl, gl, bl_))
Bq, Hq, Wq, Cq = 10, 16, 44, 576
yq, rq = torch.randn(Bq, Hq, Wq, Cq, device=dev), torch.randn(Bq, Hq, Wq, Cq, device=dev)
coefq, gateq = torch.cat([torch.ones(Cq, device=dev), torch.zeros(Cq, device=dev)]), torch.randn(Bq, Cq, device=dev)
Eq = yq.numel()
row("bottleneck output: BatchNorm apply + shortcut + ReLU -> fp32 + bf16 copies (10, 16, 44, 576)  (8 B read + 8 B written)", Eq * 16, lambda: ops.bn_apply16(yq, coefq, rq, True))
row("conv3 input: BatchNorm + ReLU + SE scale -> bf16 copies only (10, 16, 44, 576)  (4 B read + 4 B written)", Eq * 8, lambda: ops.se_scale_bn16(yq, coefq, gateq))
gq, smq, siq, dgq, dbq = torch.ones(Cq, device=dev), torch.zeros(Cq, device=dev), torch.ones(Cq, device=dev), torch.zeros(Cq, device=dev), torch.zeros(Cq, device=dev)
row("BatchNorm backward (reduce + finalize + apply) -> bf16 copies + shortcut gradient (10, 16, 44, 576)  (2 x 12 B read + 8 B written)", Eq * 32,
    lambda: ops.bn_bwd16(rq, yq, yq, gq, smq, siq, dgq, dbq, want_dres=True))
ops.set_precision("fp32")
xm = torch.randn(10, 128, 352, 64, device=dev)
row("maxpool 3x3/s2 fwd (10, 128, 352, 64)  (read x, write y + 1 B index)", xm.numel() * 4 + xm.numel() // 4 * 5, lambda: ops.maxpool3x3s2_fwd(xm))
torch.cuda.synchronize()
if args.meta:
    json.dump(meta, open(args.meta, "w"), indent=1)
