"""ResNet restatement (timm==0.5.4 ``resnet18 / resnet34 / resnet50``; PARITY UNPINNED - timm is not vendored; cross-checked against the
independent Hugging Face ``transformers`` ResNet in tests/test_oracle_pinning.py).  TEST INFRASTRUCTURE: only tests/ import this.

These are the trunks the reference's constructors DEFAULT to (transfuser.py:15: image_architecture='resnet34', lidar_architecture='resnet18');
its ImageCNN / LidarEncoder use them under timm's own attribute names (conv1, bn1, act1, maxpool, layer1..4, global_pool, fc: the branch of
transfuser.py:383-416 that needs no re-labelling).  timm specifics kept: 7x7 / stride-2 stem, 3x3 / stride-2 max pool, BasicBlock (expansion 1)
and Bottleneck (expansion 4, stride on the 3x3), downsample = 1x1 conv (stride) + BN, zero-initialised last BN gamma of every residual branch,
kaiming-normal (fan_out, relu) convolutions."""
import torch
from torch import nn


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
        sc = x
        x = self.act1(self.bn1(self.conv1(x)))
        x = self.bn2(self.conv2(x))
        if self.downsample is not None:
            sc = self.downsample(sc)
        return self.act2(x + sc)


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
        sc = x
        x = self.act1(self.bn1(self.conv1(x)))
        x = self.act2(self.bn2(self.conv2(x)))
        x = self.bn3(self.conv3(x))
        if self.downsample is not None:
            sc = self.downsample(sc)
        return self.act3(x + sc)


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
            self.add_module("layer%d" % (i + 1), nn.Sequential(*blocks))
            self.feature_info.append(dict(num_chs=prev, reduction=red, module="layer%d" % (i + 1)))
        self.num_features = prev
        self.global_pool = nn.AdaptiveAvgPool2d(1)      # timm: SelectAdaptivePool2d(pool_type='avg', flatten=True); the backbone flattens itself
        self.fc = nn.Linear(prev, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        for m in self.modules():
            if hasattr(m, "zero_init_last_bn"):
                m.zero_init_last_bn()

    def forward_features(self, x):
        x = self.maxpool(self.act1(self.bn1(self.conv1(x))))
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))

    def forward(self, x):      # timm 0.5.4 resnet.py ResNet.forward (drop_rate 0); late_fusion.py:129-132 empties global_pool / fc
        return self.fc(self.global_pool(self.forward_features(x)))


def resnet18(in_chans=3):
    return ResNet(BasicBlock, (2, 2, 2, 2), in_chans)


def resnet34(in_chans=3):
    return ResNet(BasicBlock, (3, 4, 6, 3), in_chans)


def resnet50(in_chans=3):
    return ResNet(Bottleneck, (3, 4, 6, 3), in_chans)


ARCH = {"resnet18": resnet18, "resnet34": resnet34, "resnet50": resnet50}
