"""RegNetY restatement (timm==0.5.4 ``regnety_032``; PARITY UNPINNED - timm is not vendored).

Follows SURVEY.md App. D1: RegNetCfg(w0=80, wa=42.63, wm=2.66, group_w=24, depth=21, se_ratio=.25)
-> widths [72,216,576,1512], depths [2,5,13,1].  Module / parameter names replicate timm's so
that the reference's re-labelling code (team_code_transfuser/transfuser.py:380-393, 442-488)
runs unmodified and state_dict keys match reference checkpoints.
"""
import math
import numpy as np
import torch
from torch import nn


class BatchNormAct2d(nn.BatchNorm2d):
    """BN with the activation folded in (why transfuser.py:386 sets act1 = Sequential())."""

    def __init__(self, num_features, apply_act=True):
        super().__init__(num_features, eps=1e-5, momentum=0.1)
        self.act = nn.ReLU(inplace=True) if apply_act else nn.Identity()

    def forward(self, x):
        return self.act(super().forward(x))


class ConvBnAct(nn.Module):
    def __init__(self, cin, cout, k, stride=1, groups=1, apply_act=True):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride, k // 2, groups=groups, bias=False)
        self.bn = BatchNormAct2d(cout, apply_act)

    def forward(self, x):
        return self.bn(self.conv(x))


class SEModule(nn.Module):
    def __init__(self, ch, rd):
        super().__init__()
        self.fc1 = nn.Conv2d(ch, rd, 1, bias=True)
        self.act = nn.ReLU(inplace=True)
        self.fc2 = nn.Conv2d(rd, ch, 1, bias=True)
        self.gate = nn.Sigmoid()

    def forward(self, x):
        s = x.mean((2, 3), keepdim=True)
        return x * self.gate(self.fc2(self.act(self.fc1(s))))


class Bottleneck(nn.Module):
    """Y block: 1x1 -> grouped 3x3 (stride) -> SE(rd = round(in*se_ratio)) -> 1x1 (no act) + shortcut -> ReLU."""

    def __init__(self, cin, cout, stride, group_w, se_ratio):
        super().__init__()
        self.conv1 = ConvBnAct(cin, cout, 1)
        self.conv2 = ConvBnAct(cout, cout, 3, stride, groups=cout // group_w)
        self.se = SEModule(cout, int(round(cin * se_ratio))) if se_ratio else None
        self.conv3 = ConvBnAct(cout, cout, 1, apply_act=False)
        self.act3 = nn.ReLU(inplace=True)
        self.downsample = ConvBnAct(cin, cout, 1, stride, apply_act=False) if (cin != cout or stride != 1) else None

    def forward(self, x):
        sc = x
        x = self.conv2(self.conv1(x))
        if self.se is not None:
            x = self.se(x)
        x = self.conv3(x)
        if self.downsample is not None:
            sc = self.downsample(sc)
        return self.act3(x + sc)


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
    """Design-space quantisation of the RegNet paper (generate_regnet + adjust_widths_groups_comp)."""
    ws_cont = np.arange(depth) * wa + w0
    ks = np.round(np.log(ws_cont / w0) / np.log(wm))
    ws = np.round(w0 * np.power(wm, ks) / q) * q
    widths, counts = np.unique(ws.astype(int), return_counts=True)
    widths = [int(round(w / group_w) * group_w) for w in widths]  # bottleneck ratio 1: snap to group width
    return widths, [int(c) for c in counts]


class RegNet(nn.Module):
    def __init__(self, widths, depths, group_w=24, se_ratio=0.25, in_chans=3, stem_width=32, num_classes=1000):
        super().__init__()
        self.stem = ConvBnAct(in_chans, stem_width, 3, 2)
        self.feature_info = [dict(num_chs=stem_width, reduction=2, module="stem")]
        prev, red = stem_width, 2
        for i, (w, d) in enumerate(zip(widths, depths)):
            self.add_module("s%d" % (i + 1), RegStage(prev, w, d, group_w, se_ratio))
            prev, red = w, red * 2
            self.feature_info.append(dict(num_chs=w, reduction=red, module="s%d" % (i + 1)))
        self.num_features = prev
        self.fc = nn.Linear(prev, num_classes)  # stand-in for timm's head.fc; the reference drops it
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
                m.weight.data.normal_(0, math.sqrt(2.0 / fan_out))
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, Bottleneck):
                nn.init.zeros_(m.conv3.bn.weight)  # zero_init_last_bn


def _regnet_forward(self, x):
    """timm RegNet.forward = forward_features (stem, s1..s4) + head; late_fusion.py:126-130 replaces head / fc / global_pool by identities."""
    x = self.s4(self.s3(self.s2(self.s1(self.stem(x)))))
    for name in ("global_pool", "head"):
        m = getattr(self, name, None)
        if isinstance(m, nn.Module):
            x = m(x)
    return x


RegNet.forward = _regnet_forward


def regnety_032(in_chans=3):
    widths, depths = regnet_widths(80, 42.63, 2.66, 21, 24)
    assert widths == [72, 216, 576, 1512] and depths == [2, 5, 13, 1], (widths, depths)
    return RegNet(widths, depths, 24, 0.25, in_chans)
