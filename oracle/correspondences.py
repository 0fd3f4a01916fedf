"""LiDAR <-> camera correspondences of the geometric-fusion backbone (team_code_transfuser/data.py:632-842).  TEST INFRASTRUCTURE.

``project_pairs`` restates ``lidar_bev_cam_correspondences`` (data.py:675-842) up to the point where the reference has its two lists of
pixel coordinates, with the arithmetic types NumPy >= 2 gives the reference's expressions on a float32 cloud (NEP 50: ``focal_x`` is a
float64 scalar, so the centre camera is evaluated in float64 on the widened float32 coordinates; the BEV pixel of a point is float32
arithmetic; the rotated clouds come out of ``R.dot`` in float64).  Under the NumPy 1.x of the reference's own environment the centre
camera is float32 and the BEV pixel float64 instead - the reference's result depends on its NumPy version for points within one rounding
of a pixel boundary; the version pinned here is the one that can be EXECUTED here (tests/test_oracle_pinning_data.py runs the
reference's source).  What matters downstream are the 32-pixel cells: every kept (camera, point) entry is a pair
(BEV cell (bx, by) in 8 x 8, image cell (cx, cy) in 22 x 5), in the reference's order: left camera, centre, right, each in cloud order.

``correspondences_at_one_scale`` (data.py:632-673) then gives every cell its first <= 5 partners in that order, or - more than 5 -
``random.sample(list, 5)`` from Python's GLOBAL generator: not reproducible from the inputs.  ``select`` replaces that draw by a
counter-based one with the same distribution (a uniformly random 5-subset in uniformly random order): every entry of a crowded cell gets
the priority hash32(seed, sample, list, cell, entry key) and the five smallest win, in ascending priority.  The HIP kernel
(csrc/correspond.cpp) implements exactly this, so the two can be compared for equality; against the reference itself the tests compare
the <= 5 cells exactly and the crowded cells by membership / multiplicity.
"""
import numpy as np

LIDAR_CELLS = (8, 8)
CAM_CELLS = (22, 5)


def camera_constants():
    """The reference's expressions (data.py:688-712), evaluated by NumPy in float64 exactly as there."""
    img_width, img_height, fov_width = 352, 160, 60
    fov_height = 2.0 * np.arctan((img_height / img_width) * np.tan(0.5 * np.radians(fov_width)))
    fov_height = np.rad2deg(fov_height)
    focal_x = img_width / (2.0 * np.tan(np.deg2rad(fov_width) / 2.0))
    focal_y = img_height / (2.0 * np.tan(np.deg2rad(fov_height) / 2.0))
    th_l, th_r = np.radians(-60.0), np.radians(60.0)
    return dict(focal_x=float(focal_x), focal_y=float(focal_y), cos_l=float(np.cos(th_l)), sin_l=float(np.sin(th_l)), cos_r=float(np.cos(th_r)), sin_r=float(np.sin(th_r)))


def project_pairs(world, key_stride=None):
    """world (N, >= 3) float32, CARLA frame (x left, y forward, z up).  Returns int arrays (bx, by, cx, cy, key) of the kept entries in the
    reference's order; key = camera * key_stride + point index (camera 0 = left, 1 = centre, 2 = right; key_stride = N unless the cloud is a
    prefix of a longer padded buffer - the kernel's keys count in units of the buffer length)."""
    k = camera_constants()
    w = np.asarray(world, np.float32)[:, :3].copy()
    n = key_stride or w.shape[0]
    x = -w[:, 0]                                                        # data.py:715 (float32)
    y, z = w[:, 1], (w[:, 2] + np.float32(2.5 - 2.3)).astype(np.float32)  # :721: float32 + (weak) Python float
    keep = (np.abs(x) < np.float32(16.0)) & (y < np.float32(32.0)) & (y > 0)
    x64, y64, z64 = x.astype(np.float64), y.astype(np.float64), z.astype(np.float64)
    out = []
    with np.errstate(divide="ignore", invalid="ignore"):
        for cam, (c, s) in enumerate(((k["cos_l"], k["sin_l"]), (1.0, 0.0), (k["cos_r"], k["sin_r"]))):
            if cam == 1:
                xr, yr, zr = x64, y64, z64
            else:                                                          # R.dot(lidar.T).T (:746-751, :770-775): float64, third column of R is (0, 0, 1)
                xr = (c * x64 + (-s) * y64) + 0.0 * z64
                yr = (s * x64 + c * y64) + 0.0 * z64
                zr = (0.0 * x64 + 0.0 * y64) + 1.0 * z64
            px = ((k["focal_x"] * xr) / yr) + 176.0
            py = ((k["focal_y"] * zr) / yr) + 80.0
            ok = keep & (px > 0) & (px < 352) & (py > 0) & (py < 160)
            if cam == 0:
                ok &= px >= 176.0
                px = px - 176.0
            elif cam == 1:
                px = px + 176.0
            else:
                ok &= px < 176.0
                px = px + 176.0 + 352
            idx = np.nonzero(ok)[0]
            bev_x = ((x[idx] + np.float32(16.0)) * np.float32(8)).astype(np.int64)                    # :811: float32 arithmetic, int() truncates
            bev_y = 255 - (y[idx] * np.float32(8)).astype(np.int64)                                     # :813
            img_x = px[idx].astype(np.int64)                                                              # :817
            img_y = 159 - py[idx].astype(np.int64)                                                        # :819
            # the reference indexes its 8 x 8 lists with these (a float32 sum rounding up to 256, or bev_y = -1, raises / wraps there): clamped
            bx, by = np.clip(bev_x // 32, 0, 7), np.clip(bev_y // 32, 0, 7)
            out.append(np.stack([bx, by, img_x // 32, img_y // 32, cam * n + idx], 1))
    e = np.concatenate(out, 0) if out else np.zeros((0, 5), np.int64)
    return e[:, 0], e[:, 1], e[:, 2], e[:, 3], e[:, 4]


def hash32(x):
    x = np.asarray(x, np.uint64) & 0xffffffff
    x ^= x >> 16; x = (x * 0x7feb352d) & 0xffffffff
    x ^= x >> 15; x = (x * 0x846ca68b) & 0xffffffff
    x ^= x >> 16
    return x


def priority(seed, sample, lst, cell, key):
    """32-bit priority of an entry of a crowded cell; lst 0 = per BEV cell (cell = bx * 8 + by), 1 = per image cell (cell = cx * 5 + cy)."""
    site = (sample * 2 + lst) * 128 + cell
    return hash32((np.asarray(key, np.uint64) * 0x9E3779B9 + hash32(np.uint64(seed) ^ ((np.uint64(site) * 0x85ebca6b) & 0xffffffff))) & 0xffffffff)


def select(cells, partners, keys, ncell, seed, sample, lst):
    """cells (E,) flat cell id of every entry, partners (E, 2) what the entry appends there, keys (E,) -> (ncell, 5, 2) int64."""
    out = np.zeros((ncell, 5, 2), np.int64)
    for c in range(ncell):
        m = np.nonzero(cells == c)[0]
        if m.size == 0:
            continue
        if m.size > 5:
            pr = priority(seed, sample, lst, c, keys[m])
            order = np.lexsort((keys[m], pr))[:5]
        else:
            order = np.argsort(keys[m], kind="stable")
        out[c, :order.size] = partners[m[order]]
    return out


def lidar_bev_cam_correspondences(world, seed=0, sample=0, key_stride=None):
    """-> (bev_points (8, 8, 5, 2): image cells of the points in a BEV cell, cam_points (22, 5, 5, 2): BEV cells of the points in an image cell)."""
    bx, by, cx, cy, key = project_pairs(world, key_stride)
    bev = select(bx * 8 + by, np.stack([cx, cy], 1), key, 64, seed, sample, 0).reshape(8, 8, 5, 2)
    cam = select(cx * 5 + cy, np.stack([bx, by], 1), key, 110, seed, sample, 1).reshape(22, 5, 5, 2)
    return bev, cam


def cell_lists(world):
    """The reference's intermediate lists (for the membership checks): {flat BEV cell: [(cx, cy), ...]}, {flat image cell: [(bx, by), ...]} in order."""
    bx, by, cx, cy, key = project_pairs(world)
    lb, lc = {}, {}
    for i in np.argsort(key, kind="stable"):
        lb.setdefault(int(bx[i] * 8 + by[i]), []).append((int(cx[i]), int(cy[i])))
        lc.setdefault(int(cx[i] * 5 + cy[i]), []).append((int(bx[i]), int(by[i])))
    return lb, lc
