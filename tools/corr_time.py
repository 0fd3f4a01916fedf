import sys, torch, numpy as np
sys.path.insert(0, '.')
from transfuser_amd import ops
from transfuser_amd.data import synthetic_cloud
pts = synthetic_cloud(12, 40000, 0); pts[..., 1] *= -1      # CARLA frame-ish: forward y positive
pts = torch.from_numpy(pts).cuda()
for _ in range(3): ops.lidar_cam_correspondences(pts, None, seed=1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.lidar_cam_correspondences(pts, None, seed=1)
e1.record(); e1.synchronize()
print("lidar_cam_correspondences 12 x 40000 points: %.1f us per batch" % (e0.elapsed_time(e1) * 100))
