"""Batch contract of the hot path (team_code_transfuser/data.py:103-356, train.py:246-271) and the
synthetic batch of SURVEY.md section 8(d).  File decoding / augmentation are out of scope."""
import math
import numpy as np
import torch


def synthetic_cloud(B, n_points=32768, seed=0):
    """(B, N, 4) f32 cloud: x~U(-20,20), y~U(-36,4), z~U(-4,1), intensity~U(0,1) - exercises
    out-of-range points and both height bins."""
    rng = np.random.default_rng(seed)
    pts = np.stack([rng.uniform(-20, 20, (B, n_points)), rng.uniform(-36, 4, (B, n_points)),
                    rng.uniform(-4, 1, (B, n_points)), rng.uniform(0, 1, (B, n_points))], axis=-1)
    return pts.astype(np.float32)


def draw_target_point_image(px, py, size=256, radius=5, thickness=3):
    """Ring of radius 5 / thickness 3 at pixel (px, py) (data.py:616-630 draws it with cv2.circle;
    here: |dist - radius| <= thickness/2, values in {0,1})."""
    yy, xx = np.mgrid[0:size, 0:size]
    d = np.sqrt((xx - px) ** 2 + (yy - py) ** 2)
    return (np.abs(d - radius) <= thickness / 2.0).astype(np.float32)[None]


def synthetic_batch(B, H=256, W=704, seed=0, hist_fn=None, n_points=32768):
    """Seeded synthetic training batch with the dataset's shapes/dtypes (CPU tensors).

    ``hist_fn(points (N,4) float32 ndarray) -> (2,256,256) float32`` builds the LiDAR BEV
    histogram; the caller passes the HIP op (bench / product) or the oracle (tests)."""
    g = torch.Generator().manual_seed(seed)
    rng = np.random.default_rng(seed + 1)
    cloud = synthetic_cloud(B, n_points, seed)
    lidar = np.stack([hist_fn(cloud[b]) for b in range(B)])
    tpi = np.stack([draw_target_point_image(rng.integers(8, 248), rng.integers(8, 248)) for _ in range(B)])
    label = torch.zeros(B, 20, 7)
    for b in range(B):
        k = int(rng.integers(0, 9))
        if k:
            label[b, :k, 0:2] = torch.from_numpy(rng.uniform(1, 254, (k, 2))).float()
            label[b, :k, 2:4] = torch.from_numpy(rng.uniform(8, 40, (k, 2))).float()
            label[b, :k, 4] = torch.from_numpy(rng.uniform(-math.pi, math.pi, k)).float()
            label[b, :k, 5] = torch.from_numpy(rng.uniform(0, 8, k)).float()
            label[b, :k, 6] = torch.from_numpy(rng.integers(0, 2, k)).float()
    g2 = torch.Generator().manual_seed(seed + 2)   # C4 correspondences (data.py:632-675): bev_points (x<22, y<5), cam_points (<8)
    bev_points = torch.stack((torch.randint(0, 22, (B, 8, 8, 5), generator=g2), torch.randint(0, 5, (B, 8, 8, 5), generator=g2)), -1)
    cam_points = torch.randint(0, 8, (B, 22, 5, 5, 2), generator=g2)
    return dict(
        bev_points=bev_points, cam_points=cam_points,
        rgb=torch.randint(0, 256, (B, 3, H, W), generator=g).float(),
        lidar=torch.from_numpy(lidar).float(),
        lidar_raw=torch.from_numpy(np.pad(cloud, ((0, 0), (0, 40000 - n_points), (0, 0)))),
        num_points=torch.full((B,), n_points, dtype=torch.int32),
        target_point_image=torch.from_numpy(tpi).float(),
        ego_vel=torch.rand(B, 1, generator=g) * 8,
        target_point=torch.rand(B, 2, generator=g) * 40 - 10,
        ego_waypoint=torch.rand(B, 4, 2, generator=g) * 14 - 2,
        bev=torch.randint(0, 3, (B, 160, 160), generator=g),
        label=label,
        depth=torch.rand(B, H, W, generator=g),
        semantic=torch.randint(0, 7, (B, H, W), generator=g),
    )
