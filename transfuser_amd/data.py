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


# ================================================================================================ datasets + GPU-side batch preparation
CONVERTER = [0, 0, 0, 0, 4, 0, 5, 2, 6, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 3, 3, 0, 0, 5]      # config.py:88-117 (CARLA class -> 7 training classes)


def _hist_numpy(points):
    """data.py:446-470 in its integer-exact closed form on the host: used by the synthetic DataLoader workers, which
    must not touch the GPU (train.py:159-163)."""
    x, y, z = points[:, 0], points[:, 1], points[:, 2]
    ok = (x >= -16) & (x <= 16) & (y >= -32) & (y <= 0)
    xb = np.minimum(np.floor(x[ok] * 8).astype(np.int64) + 128, 255)
    yb = np.minimum(np.floor(y[ok] * 8).astype(np.int64) + 256, 255)
    cnt = np.zeros((2, 256, 256), np.int64)
    np.add.at(cnt, ((z[ok] <= -2.3).astype(np.int64), yb, 255 - xb), 1)
    return (np.minimum(cnt, 5) / 5).astype(np.float32)


class SyntheticDataset(torch.utils.data.Dataset):
    """``n`` seeded samples with the dataset's shapes / dtypes (one item = the dict data.py:103-356 returns for one frame): the offline
    stand-in for the 210 GB CARLA dataset; sample i is reproducible from (seed, i)."""

    def __init__(self, n, H=160, W=704, seed=0, n_points=8192):
        self.n, self.H, self.W, self.seed, self.n_points = int(n), H, W, seed, n_points

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        b = synthetic_batch(1, self.H, self.W, seed=self.seed * 100003 + int(i), hist_fn=_hist_numpy, n_points=self.n_points)
        out = {k: v[0] for k, v in b.items() if k not in ("lidar_raw", "num_points")}
        out["ego_vel"] = b["ego_vel"][0]
        return out


class CARLA_Data(torch.utils.data.Dataset):
    """Reader of the reference's on-disk format (data.py:46-97): ``root`` = list of town folders, each with route folders holding
    rgb/ depth/ semantics/ topdown/encoded_*.png, lidar/*.npy (pickled (frame, points[N, 4])), label_raw/*.json, measurements/*.json.
    ``__getitem__`` only DECODES (PIL / numpy / json - the CPU part that cannot move) and returns raw arrays; alignment, histogram, crops,
    depth / class / BEV decoding and the augmentation geometry are applied to the whole collated batch on the GPU by ``GpuBatchPrep``."""

    def __init__(self, root, config, shared_dict=None):
        """``shared_dict``: the reference's ``--use_disk_cache 1`` (train.py:77-90, data.py:133-153: a diskcache.Cache under $SCRATCH holding
        the decoded sample) - here a directory path: the DECODED raw item of every index is stored there once (torch.save, atomic rename)
        and read back on later epochs / by the other ranks instead of decoding PNG / JSON / NPY from the slow storage again."""
        self.config = config
        self.cache_dir = shared_dict
        if self.cache_dir:
            os.makedirs(self.cache_dir, exist_ok=True)
        self.seq_len, self.pred_len = int(config.seq_len), int(config.pred_len)
        assert self.seq_len == 1, "seq_len 1 (the reference's only configuration, config.py:14)"
        self.frames = []
        for sub_root in root:
            for route in sorted(os.listdir(sub_root)):
                route_dir = os.path.join(sub_root, route)
                if not os.path.isdir(os.path.join(route_dir, "lidar")):
                    continue
                num_seq = len(os.listdir(os.path.join(route_dir, "lidar")))
                for seq in range(2, num_seq - self.pred_len - self.seq_len - 2):      # data.py:61: ignore the first / last two frames
                    self.frames.append((route_dir, seq))

    def __len__(self):
        return len(self.frames)

    def __getitem__(self, index):
        if not self.cache_dir:
            return self._decode(index)
        route_dir, seq = self.frames[index]
        import hashlib
        # the key covers everything that shapes a decoded item: the route, the decode-affecting config fields and a format version
        # (a changed max_lidar_points / seq_len / pred_len used to hit stale files)
        tag = "%s|v2|%d|%d|%d" % (route_dir, getattr(self.config, "max_lidar_points", 0), self.seq_len, self.pred_len)
        path = os.path.join(self.cache_dir, "%s_%04d.pt" % (hashlib.sha1(tag.encode()).hexdigest()[:16], seq))
        if os.path.exists(path):
            item = torch.load(path)
        else:
            item = self._decode(index, augment=False)
            tmp = "%s.%d.tmp" % (path, os.getpid())
            torch.save(item, tmp)
            os.replace(tmp, path)          # atomic: concurrent DataLoader workers / ranks never read a partial file
        # the augmentation draw is per access (data.py:213-220 draws it after the cache lookup), so it is re-done on the cached labels
        item.update(host_sample_geometry(item.pop("_labels"), item.pop("_meas"), self.config, self.pred_len))
        return item

    def _decode(self, index, augment=True):
        import json
        from PIL import Image
        route_dir, seq = self.frames[index]
        rd = lambda sub, name: np.asarray(Image.open(os.path.join(route_dir, sub, name)).convert("RGB"))
        with open(os.path.join(route_dir, "measurements", "%04d.json" % seq)) as f:
            meas = json.load(f)
        labels = []
        for i in range(self.seq_len + self.pred_len):
            with open(os.path.join(route_dir, "label_raw", "%04d.json" % (seq + i))) as f:
                labels.append(json.load(f))
        lidar = np.load(os.path.join(route_dir, "lidar", "%04d.npy" % seq), allow_pickle=True)[1].astype(np.float32)
        lidar[:, 1] *= -1                                                   # data.py:170
        n = min(len(lidar), int(self.config.max_lidar_points))
        pts = np.zeros((int(self.config.max_lidar_points), 4), np.float32)
        pts[:n] = lidar[:n, :4]
        sem = np.asarray(Image.open(os.path.join(route_dir, "semantics", "%04d.png" % seq)))
        sem = sem[..., 0] if sem.ndim == 3 else sem
        host = host_sample_geometry(labels, meas, self.config, self.pred_len) if augment else dict(_labels=labels, _meas=meas)
        return dict(rgb_u8=torch.from_numpy(rd("rgb", "%04d.png" % seq).copy()), depth_u8=torch.from_numpy(rd("depth", "%04d.png" % seq).copy()),
                    sem_u8=torch.from_numpy(sem.copy()[..., None]), bev_u8=torch.from_numpy(rd("topdown", "encoded_%04d.png" % seq).copy()),
                    lidar_raw=torch.from_numpy(pts), num_points=torch.tensor(n, dtype=torch.int32),
                    ego_matrix=torch.tensor(meas["ego_matrix"], dtype=torch.float64), **host)


import os  # noqa: E402


def host_sample_geometry(labels, meas, config, pred_len, rng=None):
    """The small per-sample host logic of data.py:196-204, 272-350 (Python dict walking: labels -> boxes, waypoints, command point) with the
    augmentation draw.  Returns tensors; the box / waypoint ROTATION by the drawn angle is part of it (data.py:474-497, 305-308)."""
    import random
    rng = rng or random
    degree = 0.0
    if bool(config.augment) and rng.random() > config.inv_augment_prob:
        degree = (rng.random() * 2.0 - 1.0) * config.aug_max_rotation
    rad = np.deg2rad(degree)
    T_bev = np.array([[0, -1, 16], [-1, 0, 32], [0, 0, 1]], dtype=np.float32)
    T_bev[:2, :] *= 8
    boxes, ego_id = {}, labels[0][0]['id']
    dm = np.array([[np.cos(-rad), np.sin(-rad), 0], [-np.sin(-rad), np.cos(-rad), 0], [0, 0, 1]])      # parse_labels(..., rad=-rad)
    for r in labels[0]:
        dz, dx, dy = r['extent']
        x, y, z = r['position']
        pos = (T_bev @ dm) @ np.array([x, y, 1.0]).reshape(3, 1)
        pos = np.clip(pos, 0., 255.)
        bx, by = pos[:2, 0]
        if r['num_points'] <= 1 or bx <= 0.0 or bx >= 255.0 or by <= 0.0 or by >= 255.0:
            continue
        boxes[r['id']] = np.array([bx, by, dy * 8, dx * 8, r['yaw'] - rad, r['speed'], r['brake']])
    label_pad = np.zeros((20, 7), np.float32)
    lab = np.array(list(boxes.values()))
    if lab.shape[0] > 0:
        label_pad[:min(20, lab.shape[0])] = lab[:20]
    # ego waypoints: future ego poses relative to the current one, in the virtual LiDAR frame (data.py:374-409)
    Tv = np.linalg.inv(np.array([[1, 0, 0, 1.3], [0, 1, 0, 0.0], [0, 0, 1, 2.5], [0, 0, 0, 1.0]]))
    cur = None
    mats = []
    for i, frame in enumerate(labels[:pred_len + 1]):
        m = next((np.array(o['ego_matrix']) for o in frame if o['id'] == ego_id), np.eye(4))
        if i == 0:
            cur = np.linalg.inv(m)
        else:
            mats.append((Tv @ cur @ m)[:2, 3])
    wp = np.array(mats)
    dmat = np.array([[np.cos(rad), np.sin(rad)], [-np.sin(rad), np.cos(rad)]])
    wp = (dmat @ wp.T).T
    th = meas['theta'] + rad
    R = np.array([[np.cos(np.pi / 2 + th), -np.sin(np.pi / 2 + th)], [np.sin(np.pi / 2 + th), np.cos(np.pi / 2 + th)]])
    tp = R.T.dot(np.array([meas['x_command'] - meas['x'], meas['y_command'] - meas['y']]))
    return dict(label=torch.from_numpy(label_pad), ego_waypoint=torch.from_numpy(wp.astype(np.float32)), target_point=torch.from_numpy(tp.astype(np.float32)),
                ego_vel=torch.tensor([meas['speed']], dtype=torch.float32), degree=torch.tensor(degree, dtype=torch.float32))


def align_transform(ego_matrix_0, ego_matrix_1, degree=0.0):
    """The 4x4 fp64 matrix of data.py:411-431: degree_matrix @ Tr_vehicle_to_lidar @ inv(M1) @ M0 @ Tr_lidar_to_vehicle."""
    lv = np.eye(4)
    lv[:3, :3] = np.array([[0, 1, 0], [-1, 0, 0], [0, 0, 1]], dtype=np.float32)
    lv[0, 3], lv[1, 3], lv[2, 3] = 1.3, 0.0, 2.5
    t = np.linalg.inv(lv) @ np.linalg.inv(np.asarray(ego_matrix_1, dtype=np.float64)) @ np.asarray(ego_matrix_0, dtype=np.float64) @ lv
    rad = np.deg2rad(degree)
    dmat = np.array([[np.cos(rad), np.sin(rad), 0, 0], [-np.sin(rad), np.cos(rad), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    return dmat @ t


class GpuBatchPrep:
    """Collated raw batch (``CARLA_Data`` items) -> the training batch of train.py:246-271, on the device: one H2D copy of the uint8 images
    and the float32 clouds, then five kernels (csrc/dataprep.cpp) instead of per-sample numpy in the DataLoader workers."""

    def __init__(self, config, device, correspondences=False, seed=0):
        """``correspondences``: also produce ``bev_points`` / ``cam_points`` from the raw cloud (data.py:273,319-320: the geometric-fusion
        backbone's inputs); ``seed`` keys the draw for cells with more than five points (a new draw per batch, like the reference's
        ``random.sample`` per sample)."""
        self.config, self.device = config, device
        self.lut = torch.tensor(CONVERTER + [0] * (256 - len(CONVERTER)), dtype=torch.uint8, device=device)
        self.correspondences, self.seed, self.calls = correspondences, int(seed), 0

    def __call__(self, raw):
        from . import ops
        cfg, dev = self.config, self.device
        B = raw["rgb_u8"].shape[0]
        deg = raw["degree"].to(torch.float32)
        ch, cw = cfg.img_resolution
        crop_shift = (deg.double() / 60.0 * cfg.img_width / cfg.scale)          # data.py:203: crop_shift = degree / 60 * img_width / scale
        out = {}
        for key, mode, name in (("rgb_u8", "rgb", "rgb"), ("depth_u8", "depth", "depth"), ("sem_u8", "seg", "semantic")):
            src = raw[key].to(dev, non_blocking=True)
            Hs, Ws = src.shape[1], src.shape[2]
            sx = (Ws // 2 - cw // 2) + crop_shift.to(torch.int32)               # int(): truncation toward zero, like Python's int()
            out[name] = ops.image_prep(src, (ch, cw), Hs // 2 - ch // 2, sx, mode, self.lut)
        out["bev"] = ops.bev_prep(raw["bev_u8"].to(dev, non_blocking=True), deg)
        em = raw["ego_matrix"].numpy()
        T = torch.from_numpy(np.stack([align_transform(em[b], em[b], float(deg[b])) for b in range(B)]))      # seq_len 1: frame aligned to itself + rotation
        pts = raw["lidar_raw"].to(dev, non_blocking=True)
        num = raw["num_points"].to(dev, dtype=torch.int32)
        if cfg.use_point_pillars:
            out["lidar_hist"], out["lidar"] = ops.lidar_align_hist(pts, T, num, return_aligned=True)
            out["num_points"] = num
        else:
            out["lidar"] = ops.lidar_align_hist(pts, T, num)
        if self.correspondences:      # data.py:273: on the RAW (un-aligned) cloud; the loader's buffer holds it with y negated (data.py:170)
            bp, cp = ops.lidar_cam_correspondences(pts, num, seed=self.seed + 0x9E3779B1 * self.calls, y_negated=True)
            out["bev_points"], out["cam_points"] = bp.long(), cp.long()
            self.calls += 1
        tp = raw["target_point"]
        out["target_point_image"] = torch.from_numpy(np.stack([draw_target_point_image(*_tp_pixel(tp[b].numpy())) for b in range(B)])).to(dev)
        for k in ("label", "ego_waypoint", "target_point", "ego_vel"):
            out[k] = raw[k].to(dev, non_blocking=True)
        return out


def _tp_pixel(target_point):
    """data.py:616-628: local command point -> pixel of the 256 x 256 BEV frame."""
    p = target_point.astype(np.float64).copy()
    p[1] += 1.3
    p = p * 8.0
    p[1] *= -1
    p[1] = 256 - p[1]
    p[0] += 128
    p = np.clip(p.astype(np.int32), 0, 256)
    return int(p[0]), int(p[1])


def make_datasets(root_dir, config, height=160, width=704, shared_dict=None):
    """(train_set, val_set) for train.main (train.py:148-149): 'synthetic:N' -> seeded SyntheticDataset (7/8 train, 1/8 val); a directory ->
    ``CARLA_Data(root=config.train_data)`` / ``config.val_data``, the town folders GlobalConfig enumerated from the reference layout
    root/<scenario>/<town>/<route> (config.py:209-243).  A directory whose children are town folders directly (no scenario level) is
    accepted too: its town folders then train and the first one validates."""
    if str(root_dir).startswith("synthetic"):
        n = int(str(root_dir).split(":")[1]) if ":" in str(root_dir) else 64
        return SyntheticDataset(n, height, width, seed=0), SyntheticDataset(max(1, n // 8), height, width, seed=1)
    train, val = list(getattr(config, "train_data", None) or []), list(getattr(config, "val_data", None) or [])
    has_routes = lambda d: any(os.path.isdir(os.path.join(d, r, "lidar")) for r in os.listdir(d))
    if not any(has_routes(d) for d in train):      # no scenario level: root/<town>/<route>
        towns = sorted(os.path.join(root_dir, t) for t in os.listdir(root_dir) if os.path.isdir(os.path.join(root_dir, t)))
        train, val = towns, towns[:1]
    tr, va = CARLA_Data(train, config, shared_dict), CARLA_Data(val, config, shared_dict)
    if len(tr) == 0:
        raise RuntimeError("no training frames found under %s (expected <root>/<scenario>/<town>/<route>/{rgb,lidar,label_raw,...}, config.py:209-243)" % root_dir)
    return tr, va
