"""ResNet-v1b/c backbones — module tree / state_dict of
segmentron/models/backbones/resnet.py:9-247, forward on the HIP kernels."""
import torch
import torch.nn as nn

from ... import functional as F
from ...config import cfg
from .build import BACKBONE_REGISTRY

__all__ = ["ResNetV1"]


class BasicBlockV1b(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None,
                 previous_dilation=1, norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, dilation, dilation, bias=False)
        self.bn1 = norm_layer(planes)
        self.relu = nn.ReLU(True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, previous_dilation,
                               dilation=previous_dilation, bias=False)
        self.bn2 = norm_layer(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        out = F.conv_bn(x, self.conv1, self.bn1)
        out.relu = True
        out = F.conv_bn(out, self.conv2, self.bn2)
        identity = x if self.downsample is None else \
            F.conv_bn(x, self.downsample[0], self.downsample[1])
        return F.Act(F.materialize(out, residual=identity, post_relu=True))


class BottleneckV1b(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None,
                 previous_dilation=1, norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, dilation, dilation, bias=False)
        self.bn2 = norm_layer(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, 1, bias=False)
        self.bn3 = norm_layer(planes * self.expansion)
        self.relu = nn.ReLU(True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        # identity blocks: x feeds conv1 AND the residual sum — the sum's backward parks its
        # gradient (functional.GradFork), conv1's data-gradient GEMM adds it in its store path
        # instead of autograd's element-wise add
        fork = None
        if self.downsample is None and x.bn is None and not x.relu and torch.is_grad_enabled() \
                and x.t.requires_grad:
            fork = F.GradFork()
        out = F.conv_bn(x, self.conv1, self.bn1, fork=fork)
        out.relu = True
        out = F.conv_bn(out, self.conv2, self.bn2)
        out.relu = True
        out = F.conv_bn(out, self.conv3, self.bn3)
        identity = x if self.downsample is None else \
            F.conv_bn(x, self.downsample[0], self.downsample[1])
        # bn3(conv3) + identity, then ReLU: one materialising pass (resnet.py:76-79)
        return F.Act(F.materialize(out, residual=identity, post_relu=True, fork=fork))


class ResNetV1(nn.Module):
    """Standard / deep-stem ResNet; OS 8/16 via dilated layer3/4, where the FIRST block of a
    dilated stage uses dilation/2 (resnet.py:158-165).  Returns (c1, c2, c3, c4) as materialised
    NHWC activations."""

    def __init__(self, block, layers, num_classes=1000, deep_stem=False,
                 zero_init_residual=False, norm_layer=nn.BatchNorm2d):
        os_ = cfg.MODEL.OUTPUT_STRIDE
        scale = cfg.MODEL.BACKBONE_SCALE
        if os_ == 32:
            dilations, strides = [1, 1], [2, 2]
        elif os_ == 16:
            dilations, strides = [1, 2], [2, 1]
        elif os_ == 8:
            dilations, strides = [2, 4], [1, 1]
        else:
            raise NotImplementedError
        self.inplanes = int((128 if deep_stem else 64) * scale)
        super().__init__()
        self.deep_stem = deep_stem
        if deep_stem:
            mid = int(64 * scale)
            self.conv1 = nn.Sequential(
                nn.Conv2d(3, mid, 3, 2, 1, bias=False), norm_layer(mid), nn.ReLU(True),
                nn.Conv2d(mid, mid, 3, 1, 1, bias=False), norm_layer(mid), nn.ReLU(True),
                nn.Conv2d(mid, self.inplanes, 3, 1, 1, bias=False))
        else:
            self.conv1 = nn.Conv2d(3, self.inplanes, 7, 2, 3, bias=False)
        self.bn1 = norm_layer(self.inplanes)
        self.relu = nn.ReLU(True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(block, int(64 * scale), layers[0], norm_layer=norm_layer)
        self.layer2 = self._make_layer(block, int(128 * scale), layers[1], stride=2,
                                       norm_layer=norm_layer)
        self.layer3 = self._make_layer(block, int(256 * scale), layers[2], stride=strides[0],
                                       dilation=dilations[0], norm_layer=norm_layer)
        self.layer4 = self._make_layer(block, int(512 * scale), layers[3], stride=strides[1],
                                       dilation=dilations[1], norm_layer=norm_layer,
                                       multi_grid=cfg.MODEL.DANET.MULTI_GRID,
                                       multi_dilation=cfg.MODEL.DANET.MULTI_DILATION)
        self.last_inp_channels = int(512 * block.expansion * scale)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))  # unused by segmentation, kept for the
        self.fc = nn.Linear(int(512 * block.expansion * scale), num_classes)  # state_dict
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, BottleneckV1b):
                    nn.init.constant_(m.bn3.weight, 0)
                elif isinstance(m, BasicBlockV1b):
                    nn.init.constant_(m.bn2.weight, 0)

    def _make_layer(self, block, planes, blocks, stride=1, dilation=1,
                    norm_layer=nn.BatchNorm2d, multi_grid=False, multi_dilation=None):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                norm_layer(planes * block.expansion))
        if multi_grid:
            first = multi_dilation[0]
        elif dilation in (1, 2):
            first = 1
        elif dilation == 4:
            first = 2
        else:
            raise RuntimeError("=> unknown dilation size: {}".format(dilation))
        layers = [block(self.inplanes, planes, stride, dilation=first, downsample=downsample,
                        previous_dilation=dilation, norm_layer=norm_layer)]
        self.inplanes = planes * block.expansion
        for i in range(1, blocks):
            d = multi_dilation[i % len(multi_dilation)] if multi_grid else dilation
            layers.append(block(self.inplanes, planes, dilation=d, previous_dilation=dilation,
                                norm_layer=norm_layer))
        return nn.Sequential(*layers)

    def forward(self, x):
        from ... import compute_dtype
        a = F.Act(F.image_to_nhwc(x, compute_dtype()))
        if self.deep_stem:
            s = self.conv1
            a = F.conv_bn(a, s[0], s[1])
            a.relu = True
            a = F.conv_bn(a, s[3], s[4])
            a.relu = True
            a = F.conv_bn(a, s[6], self.bn1)
        else:
            a = F.conv_bn(a, self.conv1, self.bn1)
        a.relu = True
        mp = self.maxpool
        a = F.Act(F.max_pool(a, mp.kernel_size, mp.stride, mp.padding))
        feats = []
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                a = blk(a)
            feats.append(a)
        return tuple(feats)


def _reg(name, block, layers, deep_stem=False):
    def ctor(norm_layer=nn.BatchNorm2d):
        return ResNetV1(block, layers, norm_layer=norm_layer, deep_stem=deep_stem)
    ctor.__name__ = name
    BACKBONE_REGISTRY.register(ctor)
    return ctor


resnet18 = _reg("resnet18", BasicBlockV1b, [2, 2, 2, 2])
resnet34 = _reg("resnet34", BasicBlockV1b, [3, 4, 6, 3])
resnet50 = _reg("resnet50", BottleneckV1b, [3, 4, 6, 3])
resnet101 = _reg("resnet101", BottleneckV1b, [3, 4, 23, 3])
resnet152 = _reg("resnet152", BottleneckV1b, [3, 8, 36, 3])
resnet50c = _reg("resnet50c", BottleneckV1b, [3, 4, 6, 3], True)
resnet101c = _reg("resnet101c", BottleneckV1b, [3, 4, 23, 3], True)
resnet152c = _reg("resnet152c", BottleneckV1b, [3, 8, 36, 3], True)
