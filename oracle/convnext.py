"""ConvNeXt restatement (timm==0.5.4 ``convnext_tiny / small / base``; PARITY UNPINNED - timm is not vendored; cross-checked against the
independent Hugging Face ``transformers`` ConvNext in tests/test_oracle_pinning.py).  TEST INFRASTRUCTURE: only tests/ import this.

The reference re-labels this trunk in ImageCNN / LidarEncoder (transfuser.py:395-416, 457-471): stem[0] -> conv1, stem[1] (LayerNorm2d) -> bn1,
stages[i] -> layer{i+1}, head (SelectAdaptivePool2d -> LayerNorm -> Flatten -> Dropout -> fc) -> global_pool with the norm re-created as
nn.LayerNorm((512, 1, 1)) and flatten / fc emptied.  timm specifics kept: 4x4 / stride-4 patchify stem + LayerNorm2d (eps 1e-6), stage =
[LayerNorm2d + 2x2 / stride-2 conv] (stages 1-3) + blocks, block = depthwise 7x7 (bias) -> LayerNorm (channels-last, eps 1e-6) -> Linear 4x ->
GELU (erf) -> Linear -> layer scale gamma (init 1e-6) -> + shortcut, trunc-normal(0.02) weights, zero biases, no stochastic depth (rate 0)."""
from collections import OrderedDict

import torch
from torch import nn
import torch.nn.functional as F


class LayerNorm2d(nn.LayerNorm):
    """LayerNorm over the channels of an NCHW tensor."""

    def __init__(self, num_channels, eps=1e-6):
        super().__init__(num_channels, eps=eps)

    def forward(self, x):
        return F.layer_norm(x.permute(0, 2, 3, 1), self.normalized_shape, self.weight, self.bias, self.eps).permute(0, 3, 1, 2)


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.drop = nn.Dropout(0.0)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class ConvNeXtBlock(nn.Module):
    def __init__(self, dim, ls_init_value=1e-6):
        super().__init__()
        self.conv_dw = nn.Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim)
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, 4 * dim)
        self.gamma = nn.Parameter(ls_init_value * torch.ones(dim))
        self.drop_path = nn.Identity()

    def forward(self, x):
        shortcut = x
        x = self.conv_dw(x).permute(0, 2, 3, 1)
        x = self.mlp(self.norm(x)).permute(0, 3, 1, 2)
        return x.mul(self.gamma.reshape(1, -1, 1, 1)) + shortcut


class ConvNeXtStage(nn.Module):
    def __init__(self, in_chs, out_chs, stride, depth, ls_init_value):
        super().__init__()
        if in_chs != out_chs or stride > 1:
            self.downsample = nn.Sequential(LayerNorm2d(in_chs), nn.Conv2d(in_chs, out_chs, kernel_size=stride, stride=stride))
        else:
            self.downsample = nn.Identity()
        self.blocks = nn.Sequential(*[ConvNeXtBlock(out_chs, ls_init_value) for _ in range(depth)])

    def forward(self, x):
        return self.blocks(self.downsample(x))


class ConvNeXt(nn.Module):
    def __init__(self, in_chans=3, depths=(3, 3, 9, 3), dims=(96, 192, 384, 768), patch_size=4, ls_init_value=1e-6, num_classes=1000):
        super().__init__()
        self.feature_info = []
        self.stem = nn.Sequential(nn.Conv2d(in_chans, dims[0], kernel_size=patch_size, stride=patch_size), LayerNorm2d(dims[0]))
        stages, prev, red = [], dims[0], patch_size
        for i in range(4):
            stride = 2 if i > 0 else 1
            red *= stride
            stages.append(ConvNeXtStage(prev, dims[i], stride, depths[i], ls_init_value))
            prev = dims[i]
            self.feature_info += [dict(num_chs=prev, reduction=red, module="stages.%d" % i)]
        self.stages = nn.Sequential(*stages)
        self.num_features = prev
        self.norm_pre = nn.Identity()
        self.head = nn.Sequential(OrderedDict([("global_pool", nn.AdaptiveAvgPool2d(1)), ("norm", LayerNorm2d(prev)), ("flatten", nn.Flatten(1)),
                                               ("drop", nn.Dropout(0.0)), ("fc", nn.Linear(prev, num_classes))]))
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)

    def forward_features(self, x):
        return self.stages(self.stem(x))

    def forward(self, x):      # timm 0.5.4 convnext.py ConvNeXt.forward; late_fusion.py:129-132 empties the head
        return self.head(self.norm_pre(self.forward_features(x)))


def convnext_tiny(in_chans=3):
    return ConvNeXt(in_chans, (3, 3, 9, 3), (96, 192, 384, 768))


def convnext_small(in_chans=3):
    return ConvNeXt(in_chans, (3, 3, 27, 3), (96, 192, 384, 768))


def convnext_base(in_chans=3):
    return ConvNeXt(in_chans, (3, 3, 27, 3), (128, 256, 512, 1024))


ARCH = {"convnext_tiny": convnext_tiny, "convnext_small": convnext_small, "convnext_base": convnext_base}
