"""Basic blocks with the reference's module / parameter names
(segmentron/modules/basic.py:34-77) — so state_dicts interchange — whose forward runs on the
HIP kernels over deferred-BatchNorm activations (segmentron_amd.functional.Act)."""
from collections import OrderedDict

import torch.nn as nn

from .. import functional as F

__all__ = ["SeparableConv2d", "_ConvBNReLU"]


class SeparableConv2d(nn.Module):
    """depthwise 3x3 (stride, dilation, padding=dilation) -> BN -> [ReLU] -> pointwise 1x1 -> BN
    -> [ReLU]; `relu_first` moves the single ReLU in front (basic.py:34-62).  The children keep
    the reference's OrderedDict names (`block.depthwise`, `block.bn_depth`, ...); the ReLUs are
    never run as modules — they ride in the consumers' prologues."""

    def __init__(self, inplanes, planes, kernel_size=3, stride=1, dilation=1, relu_first=True,
                 bias=False, norm_layer=nn.BatchNorm2d):
        super().__init__()
        assert kernel_size == 3 and not bias
        self.relu_first = relu_first
        layers = [
            ("depthwise", nn.Conv2d(inplanes, inplanes, kernel_size, stride=stride,
                                    padding=dilation, dilation=dilation, groups=inplanes,
                                    bias=bias)),
            ("bn_depth", norm_layer(inplanes)),
            ("pointwise", nn.Conv2d(inplanes, planes, 1, bias=bias)),
            ("bn_point", norm_layer(planes)),
        ]
        if relu_first:
            layers.insert(0, ("relu", nn.ReLU()))
        else:
            layers.insert(2, ("relu1", nn.ReLU(inplace=True)))
            layers.append(("relu2", nn.ReLU(inplace=True)))
        self.block = nn.Sequential(OrderedDict(layers))

    def forward(self, act):
        b = self.block
        if self.relu_first:
            d = F.dwconv_bn(act.with_relu(), b.depthwise, b.bn_depth)
            return F.conv_bn(d, b.pointwise, b.bn_point)
        d = F.dwconv_bn(act, b.depthwise, b.bn_depth)
        d.relu = True
        p = F.conv_bn(d, b.pointwise, b.bn_point)
        p.relu = True
        return p


class _ConvBNReLU(nn.Module):
    """conv (groups=1) -> BN -> ReLU (basic.py:65-77); returns a deferred activation."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, relu6=False, norm_layer=nn.BatchNorm2d):
        super().__init__()
        if groups != 1 or relu6:
            raise NotImplementedError("grouped / ReLU6 _ConvBNReLU (MobileNetV2) is a next-row")
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation,
                              groups, bias=False)
        self.bn = norm_layer(out_channels)
        self.relu = nn.ReLU(True)

    def forward(self, act, out=None):
        a = F.conv_bn(act, self.conv, self.bn, out=out)
        a.relu = True
        return a
