"""ResNet trunks (timm==0.5.4 ``resnet18 / resnet34 / resnet50``) for the TransFuser backbones, MI355X-native.

These are the architectures the reference's constructors DEFAULT to (team_code_transfuser/transfuser.py:15: image 'resnet34', LiDAR
'resnet18'; train.py overrides them with regnety_032) and that its ImageCNN / LidarEncoder use under timm's own attribute names - the
branch of transfuser.py:383-416 / 445-471 that needs no re-labelling.  The module tree reproduces timm's names (``conv1``, ``bn1``,
``layer1.0.conv1`` ..., ``layer2.0.downsample.0``) so state_dict keys and shapes are interchangeable with reference checkpoints; the ``nn``
containers only HOLD parameters - every forward goes through the HIP kernels (functions.StemFn with the 3x3 / s2 max pool folded in,
functions.ConvBnFn = conv + BatchNorm (+ residual) (+ ReLU) with the statistics gathered by the convolution's epilogue).  NHWC activations,
k x k weights channels_last."""
import torch
from torch import nn

from . import functions as F_


def _cb(x, conv, bn, res=None, relu=True):
    return F_.ConvBnFn.apply(x, res, conv, bn, relu, conv.weight, bn.weight, bn.bias)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.act1 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.act2 = nn.ReLU(inplace=True)
        self.downsample = downsample

    def zero_init_last_bn(self):
        nn.init.zeros_(self.bn2.weight)

    def forward(self, x):
        sc = x if self.downsample is None else _cb(x, self.downsample[0], self.downsample[1], relu=False)
        return _cb(_cb(x, self.conv1, self.bn1), self.conv2, self.bn2, res=sc)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.act1 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.act2 = nn.ReLU(inplace=True)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.act3 = nn.ReLU(inplace=True)
        self.downsample = downsample

    def zero_init_last_bn(self):
        nn.init.zeros_(self.bn3.weight)

    def forward(self, x):
        sc = x if self.downsample is None else _cb(x, self.downsample[0], self.downsample[1], relu=False)
        return _cb(_cb(_cb(x, self.conv1, self.bn1), self.conv2, self.bn2), self.conv3, self.bn3, res=sc)


class _Stage(nn.Sequential):
    def forward(self, x):
        for blk in self:
            x = blk(x)
        return x


class ResNet(nn.Module):
    def __init__(self, block, layers, in_chans=3, widths=(64, 128, 256, 512), stem_width=64, num_classes=1000):
        super().__init__()
        self.conv1 = nn.Conv2d(in_chans, stem_width, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(stem_width)
        self.act1 = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.feature_info = [dict(num_chs=stem_width, reduction=2, module="act1")]
        prev, red = stem_width, 4
        for i, (planes, n) in enumerate(zip(widths, layers)):
            stride = 1 if i == 0 else 2
            blocks = []
            for j in range(n):
                s = stride if j == 0 else 1
                ds = None
                if s != 1 or prev != planes * block.expansion:
                    ds = nn.Sequential(nn.Conv2d(prev, planes * block.expansion, 1, s, bias=False), nn.BatchNorm2d(planes * block.expansion))
                blocks.append(block(prev, planes, s, ds))
                prev = planes * block.expansion
            red *= stride
            self.add_module("layer%d" % (i + 1), _Stage(*blocks))
            self.feature_info.append(dict(num_chs=prev, reduction=red, module="layer%d" % (i + 1)))
        self.num_features = prev
        self.global_pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(prev, num_classes)
        for m in self.modules():   # timm resnet init: kaiming-normal (fan_out, relu) convolutions, zero-initialised last BN gamma of every residual branch
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        for m in self.modules():
            if hasattr(m, "zero_init_last_bn"):
                m.zero_init_last_bn()


_ARCH = {"resnet18": (BasicBlock, (2, 2, 2, 2)), "resnet34": (BasicBlock, (3, 4, 6, 3)), "resnet50": (Bottleneck, (3, 4, 6, 3))}


def register_arch(name, block, layers, widths=(64, 128, 256, 512), stem_width=64):
    """Extra variants (tests use a tiny one)."""
    _ARCH[name] = (block, layers, widths, stem_width)


def is_resnet(architecture):
    return architecture in _ARCH or architecture.startswith("resnet")


def create_model(architecture, pretrained=False, in_chans=3):
    """Stand-in for ``timm.create_model`` (transfuser.py:380,442).  pretrained=True: TRANSFUSER_PRETRAINED may point at a timm state_dict of the
    same architecture (loaded like the RegNet one); otherwise the trunk is randomly initialised with a loud warning."""
    if architecture not in _ARCH:
        raise ValueError("transfuser_amd ResNet trunks: %s, got %r" % (sorted(_ARCH), architecture))
    cfg = _ARCH[architecture]
    net = ResNet(cfg[0], cfg[1], in_chans, *cfg[2:])
    if pretrained:
        import os
        import warnings
        path = os.environ.get("TRANSFUSER_PRETRAINED", "")
        if path and os.path.exists(path):
            sd = torch.load(path, map_location="cpu")
            own = net.state_dict()
            sd = {k: (v.contiguous(memory_format=torch.channels_last) if v.dim() == 4 else v) for k, v in sd.items()
                  if k in own and own[k].shape == v.shape and not k.startswith("fc.")}
            net.load_state_dict(sd, strict=False)
        else:
            warnings.warn("create_model(%r, pretrained=True): no ImageNet weights available (set TRANSFUSER_PRETRAINED=<timm state_dict>); the trunk is "
                          "RANDOMLY initialised - the reference starts from timm's ImageNet weights (transfuser.py:380)" % architecture)
    return net
