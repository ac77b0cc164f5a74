"""Basic blocks with the reference's module / parameter names
(segmentron/modules/basic.py:34-77) — so state_dicts interchange — whose forward runs on the
HIP kernels over deferred-BatchNorm activations (segmentron_amd.functional.Act)."""
from collections import OrderedDict

import torch.nn as nn

from .. import functional as F

__all__ = ["SeparableConv2d", "_ConvBNReLU", "InvertedResidual"]


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

    def forward(self, act, fork=None):
        b = self.block
        if self.relu_first:
            d = F.dwconv_bn(act.with_relu(), b.depthwise, b.bn_depth, fork=fork)
            return F.conv_bn(d, b.pointwise, b.bn_point)
        assert fork is None
        d = F.dwconv_bn(act, b.depthwise, b.bn_depth)
        d.relu = True
        p = F.conv_bn(d, b.pointwise, b.bn_point)
        p.relu = True
        return p


class _ConvBNReLU(nn.Module):
    """conv (groups=1 or depthwise) -> BN -> ReLU / ReLU6 (basic.py:65-77); returns a deferred
    activation."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, relu6=False, norm_layer=nn.BatchNorm2d):
        super().__init__()
        if groups not in (1, in_channels) or (groups != 1 and in_channels != out_channels):
            raise NotImplementedError("only dense and depthwise convolutions are on the HIP path")
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation,
                              groups, bias=False)
        self.bn = norm_layer(out_channels)
        self.relu = nn.ReLU6(True) if relu6 else nn.ReLU(True)
        self.relu6 = relu6
        self.depthwise = groups != 1

    def forward(self, act, out=None):
        if self.depthwise:
            a = F.dwconv_bn(act, self.conv, self.bn, out=out)
        else:
            a = F.conv_bn(act, self.conv, self.bn, out=out)
        a.relu = F.RELU6 if self.relu6 else F.RELU
        return a


class InvertedResidual(nn.Module):
    """MobileNetV2 block: [1x1 expand + BN + ReLU6] -> dw3x3 + BN + ReLU6 -> 1x1 linear + BN
    (+ x) (basic.py:139-163).  The linear bottleneck's BN stays deferred, so the NEXT block's
    expansion conv folds it into its weights (csrc/fold.hip)."""

    def __init__(self, in_channels, out_channels, stride, expand_ratio, dilation=1,
                 norm_layer=nn.BatchNorm2d):
        super().__init__()
        assert stride in [1, 2]
        self.use_res_connect = stride == 1 and in_channels == out_channels
        inter = int(round(in_channels * expand_ratio))
        layers = []
        if expand_ratio != 1:
            layers.append(_ConvBNReLU(in_channels, inter, 1, relu6=True, norm_layer=norm_layer))
        layers.extend([
            _ConvBNReLU(inter, inter, 3, stride, dilation, dilation, groups=inter, relu6=True,
                        norm_layer=norm_layer),
            nn.Conv2d(inter, out_channels, 1, bias=False),
            norm_layer(out_channels)])
        self.conv = nn.Sequential(*layers)

    def forward(self, x):
        a = x
        mods = list(self.conv)
        for m in mods[:-2]:
            a = m(a)
        a = F.conv_bn(a, mods[-2], mods[-1])
        if self.use_res_connect:
            return F.Act(F.materialize(a, residual=x))
        return a
