#!/usr/bin/env python
"""Per-shape census of the MFMA-engine calls of ONE eager training step (HIP-event timed):
which GEMM / conv shapes the step spends its time in and at what TFLOP/s.  python tools/census.py [B H precision]"""
import collections
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from transfuser_amd import ops  # noqa: E402
from transfuser_amd.config import GlobalConfig  # noqa: E402
from transfuser_amd.data import synthetic_batch  # noqa: E402
from transfuser_amd.model import LidarCenterNet  # noqa: E402
from transfuser_amd.train import Engine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 10
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
PREC = sys.argv[3] if len(sys.argv) > 3 else "fp32"
dev = torch.device("cuda", 0)
cfg = GlobalConfig(); cfg.n_layer = 4; cfg.use_target_point_image = True
torch.manual_seed(0)
model = LidarCenterNet(cfg, dev, 'transFuser', 'regnety_032', 'regnety_032', use_velocity=False).train()
hist_fn = lambda pts: ops.lidar_hist(torch.from_numpy(pts).to(dev)[None])[0].cpu().numpy()
batch = {k: v.to(dev) for k, v in synthetic_batch(B, H, 704, seed=0, hist_fn=hist_fn).items()}
eng = Engine(model, cfg, precision=PREC)
for _ in range(2):
    eng.train_step(batch)
torch.cuda.synchronize()
ops.census = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); eng.train_step(batch); e1.record(); torch.cuda.synchronize()
rows = ops.census; ops.census = None
agg = collections.OrderedDict()
for kind, shape, flops, a, b in rows:
    k = (kind, shape)
    v = agg.setdefault(k, [0, 0.0, 0.0])
    v[0] += 1; v[1] += a.elapsed_time(b) * 1e3; v[2] += flops
tot_us = sum(v[1] for v in agg.values()); tot_fl = sum(v[2] for v in agg.values())
print("# precision %s" % PREC)
print("# census of one eager step B=%d H=%d: %d engine calls, %.1f ms in the engine (event-timed, incl. launch gaps), %.1f GFLOP -> %.1f TFLOP/s; step wall %.1f ms" %
      (B, H, len(rows), tot_us / 1e3, tot_fl / 1e9, tot_fl / tot_us / 1e6, e0.elapsed_time(e1)))
print("# %-12s %-44s %6s %10s %8s %8s %6s" % ("kind", "shape (m,n,k,batch | B,Hi,Wi,Cin,Cout,ks,s,g)", "calls", "total_us", "avg_us", "TFLOP/s", "pct"))
for (kind, shape), (c, us, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-14s %-44s %6d %10.1f %8.1f %8.1f %5.1f%%" % (kind, str(shape), c, us, us / c, fl / us / 1e6, 100 * us / tot_us))
