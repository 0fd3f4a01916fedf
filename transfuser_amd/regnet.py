"""RegNetY trunks of the TransFuser backbone (timm==0.5.4 ``regnety_032`` as used by
team_code_transfuser/transfuser.py:380,442), MI355X-native.

The module tree reproduces timm's names (``stem.conv``, ``stem.bn``, ``s1.b1.conv1.conv`` ...,
``se.fc1`` ...) so state_dict keys and parameter shapes are interchangeable with reference
checkpoints; the ``nn`` containers only HOLD parameters - every forward goes through the HIP
kernels (functions.StemFn / YBlockFn).  Activations are NHWC; 3x3 weights are channels_last.
"""
import math

import numpy as np
import torch
from torch import nn

from . import functions as F_


class BatchNormAct2d(nn.BatchNorm2d):
    """Parameter/buffer holder for timm's BatchNormAct2d (BN + optional ReLU, eps 1e-5, momentum 0.1)."""

    def __init__(self, num_features, apply_act=True):
        super().__init__(num_features, eps=1e-5, momentum=0.1)
        self.apply_act = apply_act

    def forward(self, x):
        raise RuntimeError("BatchNormAct2d is executed by the enclosing block's fused HIP path")


class ConvBnAct(nn.Module):
    def __init__(self, cin, cout, k, stride=1, groups=1, apply_act=True):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride, k // 2, groups=groups, bias=False)
        self.bn = BatchNormAct2d(cout, apply_act)


class SEModule(nn.Module):
    def __init__(self, ch, rd):
        super().__init__()
        self.fc1 = nn.Conv2d(ch, rd, 1, bias=True)
        self.fc2 = nn.Conv2d(rd, ch, 1, bias=True)


class Bottleneck(nn.Module):
    def __init__(self, cin, cout, stride, group_w, se_ratio):
        super().__init__()
        self.in_chs, self.out_chs, self.stride, self.groups = cin, cout, stride, cout // group_w
        self.conv1 = ConvBnAct(cin, cout, 1)
        self.conv2 = ConvBnAct(cout, cout, 3, stride, groups=self.groups)
        self.se = SEModule(cout, int(round(cin * se_ratio)))
        self.conv3 = ConvBnAct(cout, cout, 1, apply_act=False)
        self.downsample = ConvBnAct(cin, cout, 1, stride, apply_act=False) if (cin != cout or stride != 1) else None

    def forward(self, x):
        return F_.YBlockFn.apply(x, self, *self.parameters())


class RegStage(nn.Module):
    def __init__(self, cin, cout, depth, group_w, se_ratio):
        super().__init__()
        for i in range(depth):
            self.add_module("b%d" % (i + 1), Bottleneck(cin if i == 0 else cout, cout, 2 if i == 0 else 1, group_w, se_ratio))

    def forward(self, x):
        for blk in self.children():
            x = blk(x)
        return x


def regnet_widths(w0, wa, wm, depth, group_w, q=8):
    ws_cont = np.arange(depth) * wa + w0
    ks = np.round(np.log(ws_cont / w0) / np.log(wm))
    ws = np.round(w0 * np.power(wm, ks) / q) * q
    widths, counts = np.unique(ws.astype(int), return_counts=True)
    return [int(round(w / group_w) * group_w) for w in widths], [int(c) for c in counts]


class RegNet(nn.Module):
    def __init__(self, widths, depths, group_w=24, se_ratio=0.25, in_chans=3, stem_width=32):
        super().__init__()
        self.stem = ConvBnAct(in_chans, stem_width, 3, 2)
        self.feature_info = [dict(num_chs=stem_width, reduction=2, module="stem")]
        prev, red = stem_width, 2
        for i, (w, d) in enumerate(zip(widths, depths)):
            self.add_module("s%d" % (i + 1), RegStage(prev, w, d, group_w, se_ratio))
            prev, red = w, red * 2
            self.feature_info.append(dict(num_chs=w, reduction=red, module="s%d" % (i + 1)))
        self.num_features = prev
        for m in self.modules():  # timm regnet init: conv N(0, sqrt(2/fan_out)), zero-init last BN gamma
            if isinstance(m, nn.Conv2d):
                fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
                m.weight.data.normal_(0, math.sqrt(2.0 / fan_out))
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, Bottleneck):
                nn.init.zeros_(m.conv3.bn.weight)


_ARCH = {"regnety_032": dict(w0=80, wa=42.63, wm=2.66, depth=21, group_w=24, se_ratio=0.25)}


def register_arch(name, **cfg):
    """Extra RegNetY variants (tests use a tiny one); cfg = widths/depths/group_w or the w0/wa/wm/depth form."""
    _ARCH[name] = cfg


def create_model(architecture, pretrained=False, in_chans=3):
    """Stand-in for ``timm.create_model`` on the training path (transfuser.py:380,442).

    ``pretrained=True`` (the reference's ImageCNN default, transfuser.py:380) needs timm's ImageNet weights, which cannot be downloaded
    here: point ``TRANSFUSER_PRETRAINED`` at a timm ``regnety_032`` state_dict (``torch.save(timm_model.state_dict(), path)``; the key
    names are timm's, which this class reproduces) and it is loaded (LidarCenterNet.__init__ then moves the 3x3 weights to channels_last) -
    otherwise a warning says loudly that the trunk starts from random initialisation (a recipe difference vs. the reference)."""
    net = _create(architecture, in_chans)
    if pretrained:
        import os
        import warnings
        path = os.environ.get("TRANSFUSER_PRETRAINED", "")
        if path and os.path.exists(path):
            sd = torch.load(path, map_location="cpu")
            sd = {k: v for k, v in sd.items() if not k.startswith(("head.", "fc."))}
            own = net.state_dict()
            if in_chans != 3:
                sd.pop("stem.conv.weight", None)
            sd = {k: (v.contiguous(memory_format=torch.channels_last) if v.dim() == 4 else v) for k, v in sd.items() if k in own and own[k].shape == v.shape}
            missing = net.load_state_dict(sd, strict=False)
            left = [k for k in missing.missing_keys if not k.startswith("fc.") and not (in_chans != 3 and k == "stem.conv.weight")]
            if left:
                warnings.warn("pretrained RegNet checkpoint %s lacks %d tensors (e.g. %s): they keep their random initialisation" % (path, len(left), left[0]))
        else:
            warnings.warn("create_model(%r, pretrained=True): no ImageNet weights available (set TRANSFUSER_PRETRAINED=<timm regnety_032 state_dict>); "
                          "the trunk is RANDOMLY initialised - the reference starts from timm's ImageNet weights (transfuser.py:380)" % architecture)
    return net


def _create(architecture, in_chans=3):
    if architecture not in _ARCH:
        raise ValueError("transfuser_amd supports RegNetY trunks %s on the hot path (train.py:50-53 defaults), got %r" % (sorted(_ARCH), architecture))
    c = dict(_ARCH[architecture])
    if "widths" in c:
        widths, depths = c.pop("widths"), c.pop("depths")
    else:
        widths, depths = regnet_widths(c.pop("w0"), c.pop("wa"), c.pop("wm"), c.pop("depth"), c["group_w"])
    return RegNet(widths, depths, c["group_w"], c.get("se_ratio", 0.25), in_chans)
