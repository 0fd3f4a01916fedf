"""CPU restatement of LidarCenterNetHead (team_code_transfuser/model.py:33-514) and the
mmdet==2.25.0 pieces it calls (PARITY UNPINNED: mmdet/mmcv are not vendored; restated from the
published 2.25.0 sources: models/utils/gaussian_target.py, models/losses/{utils,
gaussian_focal_loss,smooth_l1_loss,cross_entropy_loss}.py).  TEST INFRASTRUCTURE.
"""
import math
import numpy as np
import torch
from torch import nn
import torch.nn.functional as F

EPS32 = float(torch.finfo(torch.float32).eps)


def weight_reduce_loss(loss, weight=None, avg_factor=None):
    """mmdet weight_reduce_loss, reduction='mean'.  NOTE ``loss * weight`` BROADCASTS (quirk Q3)."""
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return loss.mean()
    return loss.sum() / (avg_factor + EPS32)


def gaussian_focal_loss(pred, target, alpha=2.0, gamma=4.0):
    eps = 1e-12
    pos_w = target.eq(1)
    neg_w = (1 - target).pow(gamma)
    pos = -(pred + eps).log() * (1 - pred).pow(alpha) * pos_w
    neg = -(1 - pred + eps).log() * pred.pow(alpha) * neg_w
    return pos + neg


def smooth_l1(pred, target, beta=1.0):
    d = (pred - target).abs()
    return torch.where(d < beta, 0.5 * d * d / beta, d - 0.5 * beta)


def gaussian_radius(det_size, min_overlap):
    h, w = det_size
    b1 = h + w
    c1 = w * h * (1 - min_overlap) / (1 + min_overlap)
    r1 = (b1 - torch.sqrt(b1 ** 2 - 4 * c1)) / 2
    b2 = 2 * (h + w)
    c2 = (1 - min_overlap) * w * h
    r2 = (b2 - torch.sqrt(b2 ** 2 - 16 * c2)) / 8
    a3 = 4 * min_overlap
    b3 = -2 * min_overlap * (h + w)
    c3 = (min_overlap - 1) * w * h
    r3 = (b3 + torch.sqrt(b3 ** 2 - 4 * a3 * c3)) / (2 * a3)
    return min(r1, r2, r3)


def gen_gaussian_target(heatmap, center, radius):
    """max-merge a (2r+1)^2 Gaussian (sigma=(2r+1)/6, tail < eps*max zeroed) at integer centre, clipped."""
    sigma = (2 * radius + 1) / 6
    ax = torch.arange(-radius, radius + 1, dtype=heatmap.dtype)
    g = (-(ax.view(1, -1) ** 2 + ax.view(-1, 1) ** 2) / (2 * sigma * sigma)).exp()
    g[g < torch.finfo(g.dtype).eps * g.max()] = 0
    x, y = int(center[0]), int(center[1])
    H, W = heatmap.shape[:2]
    left, right = min(x, radius), min(W - x, radius + 1)
    top, bottom = min(y, radius), min(H - y, radius + 1)
    sub = heatmap[y - top:y + bottom, x - left:x + right]
    torch.max(sub, g[radius - top:radius + bottom, radius - left:radius + right], out=sub)


class LidarCenterNetHead(nn.Module):
    def __init__(self, in_channel, feat_channel, num_classes, train_cfg):
        super().__init__()
        self.num_classes = num_classes
        self.num_dir_bins = train_cfg.num_dir_bins
        self.train_cfg = train_cfg
        mk = lambda oc: nn.Sequential(nn.Conv2d(in_channel, feat_channel, 3, padding=1), nn.ReLU(inplace=True),
                                      nn.Conv2d(feat_channel, oc, 1))
        # registration order = model.py:71-78 (matters for seeded init parity)
        self.heatmap_head = mk(num_classes)
        self.wh_head = mk(2)
        self.offset_head = mk(2)
        self.yaw_class_head = mk(self.num_dir_bins)
        self.yaw_res_head = mk(1)
        self.velocity_head = mk(1)
        self.brake_head = mk(2)
        self.loss_weights = dict(center_heatmap=1.0, wh=0.1, offset=1.0, yaw_class=1.0, yaw_res=1.0, velocity=1.0, brake=1.0)

    def forward_single(self, feat):  # model.py:127-147
        return (self.heatmap_head(feat).sigmoid(), self.wh_head(feat), self.offset_head(feat), self.yaw_class_head(feat),
                self.yaw_res_head(feat), self.velocity_head(feat), self.brake_head(feat))

    def angle2class(self, angle):  # model.py:250-267 (trunc division, residual w.r.t. bin centre)
        angle = angle % (2 * np.pi)
        apc = 2 * np.pi / float(self.num_dir_bins)
        shifted = (angle + apc / 2) % (2 * np.pi)
        cls = torch.div(shifted, apc, rounding_mode="trunc")
        return cls.long(), shifted - (cls * apc + apc / 2)

    def get_targets(self, gt_bbox_all, gt_label_all, gt_ignore_all, feat_shape):  # model.py:285-374
        img_h, img_w = self.train_cfg.lidar_resolution_height, self.train_cfg.lidar_resolution_width
        bs, _, fh, fw = feat_shape
        wr, hr = float(fw / img_w), float(fh / img_h)
        z = lambda c, dt=torch.float32: torch.zeros(bs, c, fh, fw, dtype=dt)
        hm, wh, off, yc, yr, vel, br, wgt = z(self.num_classes), z(2), z(2), z(1, torch.long), z(1), z(1), z(1, torch.long), z(2)
        for b in range(bs):
            box, lab, ign = gt_bbox_all[b], gt_label_all[b], gt_ignore_all[b]
            centers = torch.cat((box[:, [0]] * wr, box[:, [1]] * wr), dim=1)  # Q2: y scaled by width ratio
            for j, ct in enumerate(centers):
                if ign[j]:
                    continue
                cxi, cyi = ct.int()
                cx, cy = ct
                sh, sw = box[j, 3] * hr, box[j, 2] * wr
                radius = max(2, int(gaussian_radius([sh, sw], min_overlap=0.1)))
                gen_gaussian_target(hm[b, lab[j].long()], [cxi, cyi], radius)
                wh[b, 0, cyi, cxi] = sw
                wh[b, 1, cyi, cxi] = sh
                c, r = self.angle2class(box[j, 4])
                yc[b, 0, cyi, cxi] = c
                yr[b, 0, cyi, cxi] = r
                vel[b, 0, cyi, cxi] = box[j, 5]
                br[b, 0, cyi, cxi] = box[j, 6].long()
                off[b, 0, cyi, cxi] = cx - cxi
                off[b, 1, cyi, cxi] = cy - cyi
                wgt[b, :, cyi, cxi] = 1
        avg_factor = max(1, hm.eq(1).sum())
        return dict(center_heatmap_target=hm, wh_target=wh, yaw_class_target=yc.squeeze(1), yaw_res_target=yr,
                    offset_target=off, velocity_target=vel, brake_target=br.squeeze(1), wh_offset_target_weight=wgt), avg_factor

    def loss(self, preds, label, gt_labels, gt_ignore):  # model.py:150-248
        hm_p, wh_p, off_p, yc_p, yr_p, vel_p, br_p = preds
        t, af = self.get_targets(label, gt_labels, gt_ignore, hm_p.shape)
        w2 = t['wh_offset_target_weight']
        w1 = w2[:, :1]
        lw = self.loss_weights
        ce = lambda p, l: F.cross_entropy(p, l, reduction='none')
        return dict(
            loss_center_heatmap=lw['center_heatmap'] * weight_reduce_loss(gaussian_focal_loss(hm_p, t['center_heatmap_target']), None, af),
            loss_wh=lw['wh'] * weight_reduce_loss((wh_p - t['wh_target']).abs(), w2, af * 2),
            loss_offset=lw['offset'] * weight_reduce_loss((off_p - t['offset_target']).abs(), w2, af * 2),
            loss_yaw_class=lw['yaw_class'] * weight_reduce_loss(ce(yc_p, t['yaw_class_target']), w1.float(), af),  # Q3 broadcast
            loss_yaw_res=lw['yaw_res'] * weight_reduce_loss(smooth_l1(yr_p, t['yaw_res_target']), w1, af),
            loss_velocity=lw['velocity'] * weight_reduce_loss((vel_p - t['velocity_target']).abs(), w1, af),
            loss_brake=lw['brake'] * weight_reduce_loss(ce(br_p, t['brake_target']), w1.float(), af),  # Q3 broadcast
        )


# ---------------------------------------------------------------- inference decode
def get_local_maximum(heat, kernel=3):
    """mmdet 2.25 core/utils/gaussian_target.py: keep the cells that equal their kernel x kernel max-pool, zero the rest."""
    pad = (kernel - 1) // 2
    hmax = F.max_pool2d(heat, kernel, stride=1, padding=pad)
    return heat * (hmax == heat).float()


def get_topk_from_heatmap(scores, k=20):
    """mmdet 2.25: top-k over (class, y, x) of every image."""
    batch, _, height, width = scores.size()
    topk_scores, topk_inds = torch.topk(scores.view(batch, -1), k)
    topk_clses = topk_inds // (height * width)
    topk_inds = topk_inds % (height * width)
    topk_ys = topk_inds // width
    topk_xs = (topk_inds % width).int().float()
    return topk_scores, topk_inds, topk_clses, topk_ys, topk_xs


def transpose_and_gather_feat(feat, ind):
    """mmdet 2.25: (B, C, H, W) -> rows of the (B, H*W, C) view selected by ind (B, k)."""
    feat = feat.permute(0, 2, 3, 1).contiguous()
    feat = feat.view(feat.size(0), -1, feat.size(3))
    return feat.gather(1, ind.unsqueeze(2).repeat(1, 1, feat.size(2)))


def class2angle(angle_cls, angle_res, num_dir_bins):
    """model.py:270-284."""
    angle_per_class = 2 * np.pi / float(num_dir_bins)
    angle = angle_cls.float() * angle_per_class + angle_res
    angle[angle > np.pi] -= 2 * np.pi
    return angle


def decode_heatmap(preds, num_dir_bins, k=100, kernel=3):
    """model.py:436-497: preds = the 7 maps of forward_single -> (B, k, 8) boxes [x, y, w, h, yaw, velocity, brake, score]."""
    hm, wh_pred, offset_pred, yaw_class_pred, yaw_res_pred, velocity_pred, brake_pred = preds
    hm = get_local_maximum(hm, kernel=kernel)
    scores, index, labels, ys, xs = get_topk_from_heatmap(hm, k=k)
    wh = transpose_and_gather_feat(wh_pred, index)
    offset = transpose_and_gather_feat(offset_pred, index)
    yaw_class = torch.argmax(transpose_and_gather_feat(yaw_class_pred, index), -1)
    yaw_res = transpose_and_gather_feat(yaw_res_pred, index)
    velocity = transpose_and_gather_feat(velocity_pred, index)[..., 0]
    brake = torch.argmax(transpose_and_gather_feat(brake_pred, index), -1)
    yaw = class2angle(yaw_class, yaw_res.squeeze(2), num_dir_bins)
    xs = xs + offset[..., 0]
    ys = ys + offset[..., 1]
    boxes = torch.stack([xs, ys, wh[..., 0], wh[..., 1], yaw, velocity, brake], dim=2)
    boxes = torch.cat((boxes, scores[..., None]), dim=-1)
    boxes[:, :, :4] *= 4.
    return boxes, labels
