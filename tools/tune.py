#!/usr/bin/env python
"""Autotune the MFMA-engine tilings for the bench workload on this GPU and save the plan cache.
   python tools/tune.py [out_path] [B] [H,H] [prec,prec] [backbone]      (cudnn.benchmark analogue; ~10 s per precision and height;
   backbone = transFuser (default) | geometric_fusion (B = 12, H = 160 only) | latentTF (B = 16); TF_RETUNE=0 keeps the plans already loaded)"""
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from transfuser_amd import ops  # noqa: E402
from transfuser_amd.config import GlobalConfig  # noqa: E402
from transfuser_amd.data import synthetic_batch  # noqa: E402
from transfuser_amd.model import LidarCenterNet  # noqa: E402
from transfuser_amd.train import Engine  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "transfuser_amd", "plans", "mi355x.txt")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 10
hs = [int(h) for h in sys.argv[3].split(",")] if len(sys.argv) > 3 else [256, 160]
precs = sys.argv[4].split(",") if len(sys.argv) > 4 else ["fp32"]     # e.g. fp32,bf16: plans are keyed by compute precision, one file holds both
backbone = sys.argv[5] if len(sys.argv) > 5 else "transFuser"
dev = torch.device("cuda", 0)
cfg = GlobalConfig(); cfg.n_layer = 4; cfg.use_target_point_image = True
torch.manual_seed(0)
model = LidarCenterNet(cfg, dev, backbone, 'regnety_032', 'regnety_032', use_velocity=False).train()
hist_fn = lambda pts: ops.lidar_hist(torch.from_numpy(pts).to(dev)[None])[0].cpu().numpy()
eng = Engine(model, cfg, autotune=False)
if os.environ.get("TF_RETUNE", "1") == "1":
    ops.L().tf_plans_clear()   # re-tune every shape from scratch (kernels changed)
for prec, H in [(p, h) for p in precs for h in hs]:
    ops.set_precision(prec)
    batch = {k: v.to(dev) for k, v in synthetic_batch(B, H, 704, seed=0, hist_fn=hist_fn).items()}
    eng.train_step(batch); torch.cuda.synchronize()
    t0 = time.time(); eng.train_step(batch); torch.cuda.synchronize(); base = time.time() - t0
    ops.autotune(True)
    t0 = time.time(); eng.train_step(batch); torch.cuda.synchronize(); tune_s = time.time() - t0
    ops.autotune(False)
    eng.train_step(batch); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        eng.train_step(batch)
    torch.cuda.synchronize(); tuned = (time.time() - t0) / 3
    print("%s H=%d: eager step %.1f ms heuristic -> %.1f ms tuned (tuning pass %.1f s, %d plans)" % (prec, H, base * 1e3, tuned * 1e3, tune_s, ops.L().tf_plans_count()), flush=True)
os.makedirs(os.path.dirname(out), exist_ok=True)
ops.plans_save(out)
print("saved", out)
