"""Shared checks of the GPU-side batch preparation against tests/golden/dataprep.npz = outputs of the reference's OWN
data.py functions (ast-extracted and executed by tests/golden/make_golden.py)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as mg  # noqa: E402


def check_dataprep(dev):
    from transfuser_amd import data as D, ops
    from transfuser_amd.config import GlobalConfig
    gold = np.load(os.path.join(HERE, "golden", "dataprep.npz"))
    raw, labels, meas = mg.dataprep_raw()
    cfg = GlobalConfig()
    cfg.img_resolution = (16, 64)
    cfg.augment = False
    B = raw["rgb_u8"].shape[0]
    host = [D.host_sample_geometry(labels[b], meas[b], cfg, 4) for b in range(B)]
    for key, gk in (("label", "dp_label"), ("ego_waypoint", "dp_ego_waypoint"), ("target_point", "dp_target_point")):
        got = torch.stack([h[key] for h in host]).double().numpy()
        assert got.shape == gold[gk].shape, (key, got.shape, gold[gk].shape)
        assert np.abs(got - gold[gk]).max() <= 1e-5 * max(1.0, np.abs(gold[gk]).max()), (key, np.abs(got - gold[gk]).max())
    assert all(float(h["degree"]) == 0.0 for h in host)
    batch = {k: torch.from_numpy(v) for k, v in raw.items()}
    batch.update({k: torch.stack([h[k] for h in host]) for k in host[0]})
    prep = D.GpuBatchPrep(cfg, torch.device(dev))
    out = prep(batch)
    assert np.array_equal(out["rgb"].cpu().numpy(), gold["dp_rgb"].astype(np.float32)) and out["rgb"].dtype == torch.float32        # crop_image_cv2
    assert np.array_equal(out["depth"].cpu().numpy(), gold["dp_depth"].astype(np.float32))                                         # get_depth, rounded like train.py:262
    assert np.array_equal(out["semantic"].cpu().numpy(), gold["dp_semantic"].astype(np.int64)) and out["semantic"].dtype == torch.int64
    assert np.array_equal(out["bev"].cpu().numpy(), gold["dp_bev"].astype(np.int64)) and set(np.unique(gold["dp_bev"])) == {0, 1, 2}
    want = np.zeros((B, 2, 256, 256), np.float32)
    want[tuple(gold["dp_lidar_idx"].astype(np.int64))] = gold["dp_lidar_val"].astype(np.float32) / 5
    assert np.array_equal(out["lidar"].cpu().numpy(), want)                                                                          # align + histogram: integer exact
    assert out["target_point_image"].shape == (B, 1, 256, 256)
    # geometric fusion (data.py:273,319-320): bev_points / cam_points from the RAW cloud of the loader's buffer (y negated there, data.py:170)
    from oracle import correspondences as oc
    prep_geo = D.GpuBatchPrep(cfg, torch.device(dev), correspondences=True, seed=11)
    og = prep_geo(batch)
    assert og["bev_points"].shape == (B, 8, 8, 5, 2) and og["cam_points"].shape == (B, 22, 5, 5, 2) and og["bev_points"].dtype == torch.int64
    for b in range(B):
        w = raw["lidar_raw"][b, :int(raw["num_points"][b]), :3].copy()
        w[:, 1] *= -1
        wb, wc = oc.lidar_bev_cam_correspondences(w, seed=11, sample=b, key_stride=raw["lidar_raw"].shape[1])
        assert np.array_equal(og["bev_points"][b].cpu().numpy(), wb) and np.array_equal(og["cam_points"][b].cpu().numpy(), wc)
    assert int(og["bev_points"][..., 0].max()) < 22 and int(og["bev_points"][..., 1].max()) < 5 and int(og["cam_points"].max()) < 8
    # augmentation geometry: the histogram of the cloud rotated by the kernel == the oracle's histogram of the host-rotated cloud; crops shift
    from oracle import hist
    deg = torch.tensor([7.5, -13.0])
    T = torch.from_numpy(np.stack([D.align_transform(raw["ego_matrix"][b], raw["ego_matrix"][b], float(deg[b])) for b in range(B)]))
    pts = torch.from_numpy(raw["lidar_raw"]).to(dev)
    got, aligned = ops.lidar_align_hist(pts, T, torch.from_numpy(raw["num_points"]).to(dev), return_aligned=True)
    for b in range(B):
        n = int(raw["num_points"][b])
        p = raw["lidar_raw"][b, :n].astype(np.float64).copy()
        h = np.concatenate([p[:, :3] * np.array([1, -1, 1]), np.ones((n, 1))], 1)
        q = (T[b].numpy() @ h.T).T
        q[:, 1] *= -1
        assert np.array_equal(got[b].cpu().numpy(), hist.lidar_hist_exact(q[:, :3]))
        assert np.abs(aligned[b, :n, :3].cpu().numpy() - q[:, :3].astype(np.float32)).max() <= 1e-5
    sx = torch.tensor([3, 9], dtype=torch.int32)
    r = ops.image_prep(torch.from_numpy(raw["rgb_u8"]).to(dev), (16, 64), 5, sx, "rgb")
    for b in range(B):
        assert np.array_equal(r[b].cpu().numpy(), np.transpose(raw["rgb_u8"][b, 5:21, int(sx[b]):int(sx[b]) + 64], (2, 0, 1)).astype(np.float32))
    # BEV rotation: 0 degrees through the rotating branch's neighbourhood == identity; a 90 degree turn == rot90 of the shifted map
    enc = torch.from_numpy(raw["bev_u8"]).to(dev)
    assert torch.equal(ops.bev_prep(enc, torch.zeros(B)), ops.bev_prep(enc, None))
    b90 = ops.bev_prep(enc, torch.full((B,), 90.0)).cpu().numpy()
    for b in range(B):
        c = raw["bev_u8"][b][..., 2]
        c0, c1 = ((c >> 7) & 1).astype(np.float32), ((c >> 6) & 1).astype(np.float32)
        sh0, sh1 = np.zeros_like(c0), np.zeros_like(c1)
        sh0[7:], sh1[7:] = c0[:-7], c1[:-7]
        lab = np.where(sh1 > 0, 2, np.where(sh0 > 0, 1, 0))
        # skimage.transform.rotate(angle=90) turns the image counter-clockwise = np.rot90(k=1); S is even so the centre is a pixel corner
        rot = np.rot90(lab, 1)
        agree = (b90[b] == rot[90:250, 170:330]).mean()
        assert agree > 0.999, agree
