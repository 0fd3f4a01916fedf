"""CPU restatement of the PointPillars front-end (team_code_transfuser/point_pillar.py:11-122).  TEST INFRASTRUCTURE.

``torch_scatter`` (un-vendored, README.md:63) is restated with index_add_ / scatter_reduce per its published semantics
(oracle/scatter_shim has the same restatement as an importable module so the reference file itself can be imported for
pinning): parity for torch_scatter itself is unpinned.  Everything else follows the reference line by line.
"""
import torch
from torch import nn


def scatter_mean(src, index, n):
    """torch_scatter.scatter_mean(src, index, dim=0): segment sum / max(count, 1)."""
    tot = torch.zeros(n, src.shape[1], dtype=src.dtype).index_add_(0, index, src)
    cnt = torch.zeros(n, dtype=src.dtype).index_add_(0, index, torch.ones(index.shape[0], dtype=src.dtype)).clamp_(min=1)
    return tot / cnt[:, None]


def scatter_max(src, index, n):
    """torch_scatter.scatter_max(src, index, dim=0)[0]; differentiable: the gradient goes to the rows equal to the max
    (amax splits it evenly between exact ties - only exact duplicate points tie at a positive value)."""
    idx = index[:, None].expand_as(src)
    return torch.zeros(n, src.shape[1], dtype=src.dtype).scatter_reduce(0, idx, src, reduce="amax", include_self=False)


class DynamicPointNet(nn.Module):
    def __init__(self, num_input=9, num_features=(32, 32)):   # point_pillar.py:11-25
        super().__init__()
        L = []
        for nf in num_features:
            L += [nn.Linear(num_input, nf), nn.BatchNorm1d(nf), nn.ReLU(inplace=True)]
            num_input = nf
        self.net = nn.Sequential(*L)

    def forward(self, points, inverse_indices, n_pillars):     # :27-34
        return scatter_max(self.net(points), inverse_indices, n_pillars)


class PointPillarNet(nn.Module):
    def __init__(self, num_input=9, num_features=(32, 32), min_x=-10, max_x=70, min_y=-40, max_y=40, pixels_per_meter=4):   # :37-52
        super().__init__()
        self.point_net = DynamicPointNet(num_input, list(num_features))
        self.nx = (max_x - min_x) * pixels_per_meter
        self.ny = (max_y - min_y) * pixels_per_meter
        self.min_x, self.min_y, self.max_x, self.max_y = min_x, min_y, max_x, max_y
        self.pixels_per_meter = pixels_per_meter

    def index(self, lidar_list, num_points):
        """:98-117 under no_grad: filtered points, (b, x_idx, y_idx) rows, sorted unique rows, inverse indices."""
        coords, kept = [], []
        for b, points in enumerate(lidar_list):
            points = points[:num_points[b]]
            keep = (points[:, 0] >= self.min_x) & (points[:, 0] < self.max_x) & (points[:, 1] >= self.min_y) & (points[:, 1] < self.max_y)   # :70-72
            points = points[keep, :]
            c = ((points[:, [0, 1]] - torch.tensor([self.min_x, self.min_y])) * self.pixels_per_meter).long()                            # :81-83
            coords.append(torch.nn.functional.pad(c, (1, 0), mode='constant', value=b))
            kept.append(points)
        coords, kept = torch.cat(coords, 0), torch.cat(kept, 0)
        unique_coords, inverse = coords.unique(return_inverse=True, dim=0)                                                              # :88
        return kept, unique_coords, inverse

    def decorate(self, points, unique_coords, inverse):       # :54-67 (quirk Q15: x_centers from column 2, y_centers from column 1)
        dtype = points.dtype
        x_centers = unique_coords[inverse][:, 2:3].to(dtype) / self.pixels_per_meter + self.min_x
        y_centers = unique_coords[inverse][:, 1:2].to(dtype) / self.pixels_per_meter + self.min_y
        xyz = points[:, :3]
        cluster = xyz - scatter_mean(xyz, inverse, unique_coords.shape[0])[inverse]
        return torch.cat([points, cluster, xyz[:, :1] - x_centers, xyz[:, 1:2] - y_centers], dim=-1)

    def scatter_points(self, features, coords, batch_size):   # :92-96
        canvas = torch.zeros(batch_size, features.shape[1], self.ny, self.nx, dtype=features.dtype)
        canvas[coords[:, 0], :, torch.clamp(self.ny - 1 - coords[:, 1], 0, self.ny - 1), torch.clamp(coords[:, 2], 0, self.nx - 1)] = features
        return canvas

    def forward(self, lidar_list, num_points):                 # :98-122
        with torch.no_grad():
            kept, unique_coords, inverse = self.index(lidar_list, num_points)
            decorated = self.decorate(kept, unique_coords, inverse)
        features = self.point_net(decorated, inverse, unique_coords.shape[0])
        return self.scatter_points(features, unique_coords, len(lidar_list))
