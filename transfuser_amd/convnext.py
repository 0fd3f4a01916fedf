"""ConvNeXt trunks (timm==0.5.4 ``convnext_tiny / small / base``) for the TransFuser backbones, MI355X-native: the re-labelling branch of the
reference's ImageCNN / LidarEncoder (team_code_transfuser/transfuser.py:395-416, 457-471).

The module tree reproduces timm's names (``stem.0`` / ``stem.1``, ``stages.i.downsample.0/1``, ``stages.i.blocks.j.conv_dw / norm / mlp.fc1 /
mlp.fc2 / gamma``, ``head.norm``) so state_dict keys and shapes are interchangeable with reference checkpoints (incl. the aliased duplicates the
re-labelling creates); the ``nn`` containers only HOLD parameters - every forward goes through the HIP kernels (functions.CnxStemFn / CnxDownFn /
CnxBlockFn / PoolNormFn: depthwise 7x7, LayerNorm over NHWC rows, the two Linear GEMMs, exact GELU, layer scale + shortcut).  NHWC activations."""
from collections import OrderedDict

import torch
from torch import nn

from . import functions as F_


class LayerNorm2d(nn.LayerNorm):
    def __init__(self, num_channels, eps=1e-6):
        super().__init__(num_channels, eps=eps)


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.drop = nn.Dropout(0.0)
        self.fc2 = nn.Linear(hidden, dim)


class ConvNeXtBlock(nn.Module):
    def __init__(self, dim, ls_init_value=1e-6):
        super().__init__()
        self.conv_dw = nn.Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim)
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, 4 * dim)
        self.gamma = nn.Parameter(ls_init_value * torch.ones(dim))
        self.drop_path = nn.Identity()

    def forward(self, x):
        return F_.CnxBlockFn.apply(x, self, *self.parameters())


class _Blocks(nn.Sequential):
    def forward(self, x):
        for blk in self:
            x = blk(x)
        return x


class ConvNeXtStage(nn.Module):
    def __init__(self, in_chs, out_chs, stride, depth, ls_init_value):
        super().__init__()
        if in_chs != out_chs or stride > 1:
            self.downsample = nn.Sequential(LayerNorm2d(in_chs), nn.Conv2d(in_chs, out_chs, kernel_size=stride, stride=stride))
        else:
            self.downsample = nn.Identity()
        self.blocks = _Blocks(*[ConvNeXtBlock(out_chs, ls_init_value) for _ in range(depth)])

    def forward(self, x):
        if not isinstance(self.downsample, nn.Identity):
            ds = self.downsample
            x = F_.CnxDownFn.apply(x, ds, ds[0].weight, ds[0].bias, ds[1].weight, ds[1].bias)
        return self.blocks(x)


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
        for m in self.modules():   # timm convnext init: trunc-normal(0.02) conv / linear weights, zero biases
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)


_ARCH = {"convnext_tiny": ((3, 3, 9, 3), (96, 192, 384, 768)), "convnext_small": ((3, 3, 27, 3), (96, 192, 384, 768)),
         "convnext_base": ((3, 3, 27, 3), (128, 256, 512, 1024))}


def register_arch(name, depths, dims):
    """Extra variants (tests use a tiny one)."""
    _ARCH[name] = (depths, dims)


def is_convnext(architecture):
    return architecture in _ARCH or architecture.startswith("convnext")


def create_model(architecture, pretrained=False, in_chans=3):
    if architecture not in _ARCH:
        raise ValueError("transfuser_amd ConvNeXt trunks: %s, got %r" % (sorted(_ARCH), architecture))
    depths, dims = _ARCH[architecture]
    net = ConvNeXt(in_chans, depths, dims)
    if pretrained:
        import os
        import warnings
        path = os.environ.get("TRANSFUSER_PRETRAINED", "")
        if path and os.path.exists(path):
            sd = torch.load(path, map_location="cpu")
            own = net.state_dict()
            net.load_state_dict({k: v for k, v in sd.items() if k in own and own[k].shape == v.shape and not k.startswith("head.fc.")}, strict=False)
        else:
            warnings.warn("create_model(%r, pretrained=True): no ImageNet weights available (set TRANSFUSER_PRETRAINED=<timm state_dict>); the trunk is "
                          "RANDOMLY initialised - the reference starts from timm's ImageNet weights (transfuser.py:380)" % architecture)
    return net


def relabel(net, out_features, lidar_in_channels=None):
    """transfuser.py:395-416 (ImageCNN) / 457-471 + 473-490 (LidarEncoder), restated on this tree."""
    net.fc = None
    net.conv1 = net.stem._modules['0']
    net.bn1 = net.stem._modules['1']
    net.act1 = nn.Sequential()
    net.maxpool = nn.Sequential()
    for i in range(4):
        setattr(net, "layer%d" % (i + 1), net.stages._modules[str(i)])
    net.global_pool = net.head
    net.global_pool.flatten = nn.Sequential()
    net.global_pool.fc = nn.Sequential()
    net.head = nn.Sequential()
    if lidar_in_channels is None:       # ConvNeXt has no stem entry in feature_info (transfuser.py:407-411)
        net.feature_info.append(net.feature_info[3])
        net.feature_info[3] = net.feature_info[2]
        net.feature_info[2] = net.feature_info[1]
        net.feature_info[1] = net.feature_info[0]
    tmp = net.global_pool.norm
    net.global_pool.norm = nn.LayerNorm((out_features, 1, 1), tmp.eps, tmp.elementwise_affine)
    if lidar_in_channels is not None:
        old = net.conv1
        net.conv1 = nn.Conv2d(lidar_in_channels, old.out_channels, kernel_size=old.kernel_size, stride=old.stride, padding=old.padding, bias=True)
        del net.stem._modules['0']
        net.conv1.bias = old.bias
    return net
