"""mmdet.models: HEADS registry (decorator only) and build_loss for the four loss types of model.py:58-64."""
import os
import sys

import torch
import torch.nn.functional as F
from torch import nn

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "..", ".."))
from oracle import centernet as _c  # noqa: E402


class _Registry:
    def register_module(self, *a, **k):
        return lambda cls: cls


HEADS = _Registry()


class _Loss(nn.Module):
    """Common forward(pred, target, weight=None, avg_factor=None) of mmdet 2.25 losses, reduction='mean'."""

    def __init__(self, loss_weight=1.0, reduction='mean', **kw):
        super().__init__()
        assert reduction == 'mean'
        self.loss_weight = loss_weight
        self.kw = kw

    def elementwise(self, pred, target):
        raise NotImplementedError

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'mean')
        return self.loss_weight * _c.weight_reduce_loss(self.elementwise(pred, target), weight, avg_factor)


class GaussianFocalLoss(_Loss):      # losses/gaussian_focal_loss.py, alpha=2, gamma=4
    def elementwise(self, pred, target):
        return _c.gaussian_focal_loss(pred, target, self.kw.get('alpha', 2.0), self.kw.get('gamma', 4.0))


class L1Loss(_Loss):                 # losses/smooth_l1_loss.py:l1_loss
    def elementwise(self, pred, target):
        if target.numel() == 0:
            return pred.sum() * 0
        return torch.abs(pred - target)


class SmoothL1Loss(_Loss):           # losses/smooth_l1_loss.py:smooth_l1_loss, beta=1
    def elementwise(self, pred, target):
        return _c.smooth_l1(pred, target, self.kw.get('beta', 1.0))


class CrossEntropyLoss(_Loss):       # losses/cross_entropy_loss.py:cross_entropy (use_sigmoid=False, use_mask=False, ignore_index -100)
    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None, ignore_index=None):
        loss = F.cross_entropy(cls_score, label, weight=None, reduction='none', ignore_index=-100)
        if weight is not None:
            weight = weight.float()
        return self.loss_weight * _c.weight_reduce_loss(loss, weight, avg_factor)


_LOSSES = dict(GaussianFocalLoss=GaussianFocalLoss, L1Loss=L1Loss, SmoothL1Loss=SmoothL1Loss, CrossEntropyLoss=CrossEntropyLoss)


def build_loss(cfg):
    cfg = dict(cfg)
    return _LOSSES[cfg.pop('type')](**cfg)
