"""MobileNetV2 backbone — module tree / state_dict of
segmentron/models/backbones/mobilenet.py:55-153, forward on the HIP kernels."""
import torch.nn as nn

from ... import functional as F
from ...config import cfg
from ...modules import InvertedResidual, _ConvBNReLU
from .build import BACKBONE_REGISTRY

__all__ = ["MobileNetV2"]


class MobileNetV2(nn.Module):
    """Returns c1 (24 @/4), c2 (32 @/8), c3 (96 @/16), c4 (320 @/16).  Quirk kept from the
    reference: in a dilated group only the FIRST block gets the dilation (mobilenet.py:125 vs
    :128)."""

    def __init__(self, num_classes=1000, norm_layer=nn.BatchNorm2d):
        super().__init__()
        os_ = cfg.MODEL.OUTPUT_STRIDE
        self.multiplier = cfg.MODEL.BACKBONE_SCALE
        if os_ == 32:
            dilations = [1, 1]
        elif os_ == 16:
            dilations = [1, 2]
        elif os_ == 8:
            dilations = [2, 4]
        else:
            raise NotImplementedError
        setting = [[1, 16, 1, 1], [6, 24, 2, 2], [6, 32, 3, 2], [6, 64, 4, 2], [6, 96, 3, 1],
                   [6, 160, 3, 2], [6, 320, 1, 1]]
        input_channels = int(32 * self.multiplier) if self.multiplier > 1.0 else 32
        self.conv1 = _ConvBNReLU(3, input_channels, 3, 2, 1, relu6=True, norm_layer=norm_layer)
        self.planes = input_channels
        self.block1 = self._make_layer(setting[0:1], norm_layer=norm_layer)
        self.block2 = self._make_layer(setting[1:2], norm_layer=norm_layer)
        self.block3 = self._make_layer(setting[2:3], norm_layer=norm_layer)
        self.block4 = self._make_layer(setting[3:5], dilations[0], norm_layer=norm_layer)
        self.block5 = self._make_layer(setting[5:], dilations[1], norm_layer=norm_layer)
        self.last_inp_channels = self.planes
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def _make_layer(self, setting, dilation=1, norm_layer=nn.BatchNorm2d):
        feats, planes = [], self.planes
        for t, c, n, s in setting:
            oc = int(c * self.multiplier)
            stride = s if dilation == 1 else 1
            feats.append(InvertedResidual(planes, oc, stride, t, dilation, norm_layer))
            planes = oc
            for _ in range(n - 1):
                feats.append(InvertedResidual(planes, oc, 1, t, norm_layer=norm_layer))
        self.planes = planes
        return nn.Sequential(*feats)

    def forward(self, x):
        from ... import compute_dtype
        a = self.conv1(F.Act(F.image_to_nhwc(x, compute_dtype())))
        outs = []
        for i, blk in enumerate((self.block1, self.block2, self.block3, self.block4, self.block5)):
            for m in blk:
                a = m(a)
            if i > 0:
                outs.append(a)
        return tuple(outs)


@BACKBONE_REGISTRY.register()
def mobilenet_v2(norm_layer=nn.BatchNorm2d):
    return MobileNetV2(norm_layer=norm_layer)
