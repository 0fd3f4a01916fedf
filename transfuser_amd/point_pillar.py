"""PointPillars front-end (``--use_point_pillars 1``), MI355X-native drop-in for team_code_transfuser/point_pillar.py
(``DynamicPointNet`` :11-34, ``PointPillarNet`` :37-122) plus the rot90 / target-point concat of model.py:736-742.

Same parameter names (``point_net.net.{0,1,3,4}.*``).  The reference sorts (batch, x_idx, y_idx) rows with ``torch.unique`` and
uses torch_scatter; here the pillar ids come from an occupancy-grid scan (kernels in csrc/pillars.cpp, integer-exact, no sort),
the two Linear+BatchNorm1d+ReLU layers run on the MFMA GEMM engine / BN kernels, and scatter-max + canvas scatter + rot90 +
channel concat are fused into two kernels that write the LiDAR stem's NHWC input directly.
"""
import torch
from torch import nn

from . import functions as F_
from . import ops


class DynamicPointNet(nn.Module):
    def __init__(self, num_input=9, num_features=(32, 32)):
        super().__init__()
        L = []
        for nf in num_features:
            L += [nn.Linear(num_input, nf), nn.BatchNorm1d(nf), nn.ReLU(inplace=True)]
            num_input = nf
        self.net = nn.Sequential(*L)


@F_.routes_param_grads
class PillarFn(torch.autograd.Function):
    """points (B, Nmax, 4) + num_points -> NHWC (B, nx, ny, C [+ extra channels]) = rot90(PointPillarNet(points), -1) ++ extra."""

    @staticmethod
    def forward(ctx, points, num_points, extra, net, *params):
        seq = net.point_net.net
        # static shapes (train.Engine under a hipGraph): no host read of the two counts - capacity-sized buffers, the BatchNorm1d layers take their row
        # count from the device (ops.bn_rows_dev_*).  Needs plain local train-mode BatchNorm layers whose width divides 256.
        static = bool(getattr(net, "static_shapes", False)) and all(
            type(seq[li + 1]) is nn.BatchNorm1d and seq[li + 1].training and getattr(seq[li + 1], "_sync_group", None) is None and 256 % seq[li + 1].num_features == 0
            for li in range(0, len(seq), 3))
        ix = ops.pillar_index(points, num_points, net.min_x, net.max_x, net.min_y, net.max_y, net.pixels_per_meter, static=static)
        acts, x = [], ix["feat"]
        for li in range(0, len(seq), 3):
            lin, bn = seq[li], seq[li + 1]
            h = ops.linear_fwd(x, lin.weight, lin.bias)
            if static:
                z, sm, si = ops.bn_rows_dev_fwd(h, ix["totals"], bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, relu=True)
                st = (sm, si, "rows_dev")
            else:
                z, st = F_._bn(h, bn, relu=True)
            acts.append((x, h, z, st))
            x = z
        pf, arg = ops.pillar_scatter_max(x, ix["inv"], ix["P"])
        out, owner = ops.pillar_canvas(pf, ix["cellkey"], points.shape[0], ix["nx"], ix["ny"], ix["GX"], ix["GY"], extra)
        ctx.saved = (net, ix, acts, arg, owner, pf.shape[1])
        return out

    @staticmethod
    def backward(ctx, dout):
        net, ix, acts, arg, owner, C = ctx.saved
        seq = net.point_net.net
        dz = ops.pillar_canvas_bwd(dout.contiguous(), owner, ix["cellkey"], ix["inv"], arg, C, ix["GX"], ix["GY"])
        for k in range(len(acts) - 1, -1, -1):
            lin, bn = seq[3 * k], seq[3 * k + 1]
            x, h, z, st = acts[k]
            if len(st) == 3 and st[2] == "rows_dev":
                dh = ops.bn_rows_dev_bwd(dz, z, h, ix["totals"], bn.weight, st[0], st[1], F_.gbuf(bn.weight), F_.gbuf(bn.bias))
            else:
                dh, _ = F_._bn_bwd(dz, z, h, bn, st)
            ops.linear_wgrad(dh, x, F_.gbuf(lin.weight))
            F_.bias_grad(dh, lin.bias)
            if k:
                dz = ops.linear_dgrad(dh, lin.weight)
        ctx.saved = None
        return (None,) * (4 + len(ctx.needs_input_grad) - 4)


@F_.routes_param_grads
class PillarStemFn(torch.autograd.Function):
    """LiDAR stem (conv3x3/s2 without bias + BatchNormAct2d, transfuser.py:140-143) on the NHWC pillar canvas; unlike StemFn the
    input carries a gradient (into the point net)."""

    @staticmethod
    def forward(ctx, x, stem, w, gamma, beta):
        y = ops.conv_fwd(x, w, None, 2, 1, 1)
        z, st = F_._bn(y, stem.bn, relu=True)
        ctx.saved = (x, stem, w, y, z, st)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, stem, w, y, z, st = ctx.saved
        dy, _ = F_._bn_bwd(dz.contiguous(), z, y, stem.bn, st)
        ops.conv_wgrad(dy, x, F_.gbuf(w), 2, 1, 1)
        dx = ops.conv_dgrad(dy, w, x.shape, 2, 1, 1)
        ctx.saved = None
        return dx, None, None, None, None


class PointPillarNet(nn.Module):
    def __init__(self, num_input=9, num_features=(32, 32), min_x=-10, max_x=70, min_y=-40, max_y=40, pixels_per_meter=4):
        super().__init__()
        assert num_input == 9, "decorate() produces 4 + 3 + 2 features (point_pillar.py:54-67)"
        self.point_net = DynamicPointNet(num_input, list(num_features))
        self.nx = (max_x - min_x) * pixels_per_meter
        self.ny = (max_y - min_y) * pixels_per_meter
        self.min_x, self.min_y, self.max_x, self.max_y = min_x, min_y, max_x, max_y
        self.pixels_per_meter = pixels_per_meter

    def forward_nhwc(self, lidar_list, num_points, extra=None):
        """(B, nx, ny, C + Ce) NHWC: already rotated (model.py:738) and with ``extra`` (B, Ce, nx, ny) appended (model.py:742)."""
        pts = lidar_list if torch.is_tensor(lidar_list) else torch.stack(list(lidar_list))
        assert pts.dim() == 3 and pts.shape[2] == 4 and pts.dtype == torch.float32, "lidar_raw must be (B, N, 4) float32"
        return PillarFn.apply(pts.contiguous(), num_points.to(torch.int32), extra, self, *self.parameters())

    def forward(self, lidar_list, num_points):
        """Reference API (point_pillar.py:98-122): the un-rotated NCHW canvas (B, C, ny, nx)."""
        out = self.forward_nhwc(lidar_list, num_points)            # = rot90(canvas, -1) in NHWC
        return torch.rot90(out.permute(0, 3, 1, 2), 1, dims=(2, 3))
