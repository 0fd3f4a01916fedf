#!/usr/bin/env python
"""Which ATen ops (i.e. NOT our HIP kernels) still run during one eager training step, and how often."""
import os, sys, collections
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from transfuser_amd import ops
from transfuser_amd.config import GlobalConfig
from transfuser_amd.data import synthetic_batch
from transfuser_amd.model import LidarCenterNet
from transfuser_amd.train import Engine
dev = torch.device("cuda", 0)
cfg = GlobalConfig(); cfg.n_layer = 4; cfg.use_target_point_image = True
model = LidarCenterNet(cfg, dev, 'transFuser', 'regnety_032', 'regnety_032', use_velocity=False).train()
hist_fn = lambda pts: ops.lidar_hist(torch.from_numpy(pts).to(dev)[None])[0].cpu().numpy()
batch = {k: v.to(dev) for k, v in synthetic_batch(4, 160, 704, seed=0, hist_fn=hist_fn).items()}
eng = Engine(model, cfg)
for _ in range(2):
    eng.train_step(batch)
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True) as prof:
    eng.train_step(batch); torch.cuda.synchronize()
cnt = collections.Counter(); cuda_us = collections.Counter(); stacks = collections.defaultdict(collections.Counter)
for e in prof.events():
    if e.name.startswith("aten::"):
        cnt[e.name] += 1; cuda_us[e.name] += e.device_time_total if hasattr(e, "device_time_total") else 0
        if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy") and e.stack:
            fr = [s for s in e.stack if "transfuser_amd" in s or "autograd" in s][:2]
            stacks[e.name][" <- ".join(fr)] += 1
for k, v in cnt.most_common(25):
    print("%-40s %6d calls  %10.1f us device" % (k, v, cuda_us[k]))
for k, c in stacks.items():
    print("==", k)
    for s, n in c.most_common(8):
        print("   %5d  %s" % (n, s[:220]))
