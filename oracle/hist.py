"""H1: LiDAR 2-bin height histogram (team_code_transfuser/data.py:446-470).  TEST INFRASTRUCTURE.

``lidar_to_histogram_features`` follows the reference line by line on np.histogramdd;
``lidar_hist_exact`` is the integer-exact closed form the HIP
kernel implements; tests pin the two against each other.
"""
import numpy as np


def lidar_to_histogram_features(lidar):
    def splat(pc):
        xb = np.linspace(-16, 16, 32 * 8 + 1)
        yb = np.linspace(-32, 0, 32 * 8 + 1)
        h = np.histogramdd(pc[..., :2], bins=(xb, yb))[0]
        h[h > 5] = 5
        return h / 5
    below = lidar[lidar[..., 2] <= -2.3]
    above = lidar[lidar[..., 2] > -2.3]
    f = np.stack([splat(above), splat(below)], axis=-1)
    f = np.transpose(f, (2, 0, 1)).astype(np.float32)
    return np.rot90(f, -1, axes=(1, 2)).copy()


def lidar_hist_exact(lidar):
    """xbin=floor(8x)+128 (x==16 -> 255), ybin=floor(8y)+256 (y==0 -> 255); out[c, ybin, 255-xbin] = min(cnt,5)/5."""
    x, y, z = lidar[:, 0], lidar[:, 1], lidar[:, 2]
    ok = (x >= -16) & (x <= 16) & (y >= -32) & (y <= 0)
    xb = np.minimum(np.floor(x[ok] * 8).astype(np.int64) + 128, 255)
    yb = np.minimum(np.floor(y[ok] * 8).astype(np.int64) + 256, 255)
    c = (z[ok] <= -2.3).astype(np.int64)
    cnt = np.zeros((2, 256, 256), np.int64)
    np.add.at(cnt, (c, yb, 255 - xb), 1)
    return (np.minimum(cnt, 5) / 5).astype(np.float32)
