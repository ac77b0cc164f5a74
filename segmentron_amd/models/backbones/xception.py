"""Xception65 backbone — module tree / state_dict of
segmentron/models/backbones/xception.py:10-165, forward on HIP kernels."""
import torch
import torch.nn as nn

from ... import functional as F
from ...config import cfg
from ...modules import SeparableConv2d
from .build import BACKBONE_REGISTRY

__all__ = ["Xception65", "XceptionBlock", "xception65"]


class XceptionBlock(nn.Module):
    """Three separable convs + {conv, sum, none} skip (xception.py:10-51).  The residual add is
    the one place a block output is materialised: out = BN(pw3_raw) + shortcut in ONE pass
    (reference: bn kernel + bn kernel + add kernel)."""

    def __init__(self, channel_list, stride=1, dilation=1, skip_connection_type="conv",
                 relu_first=True, low_feat=False, norm_layer=nn.BatchNorm2d):
        super().__init__()
        assert len(channel_list) == 4
        if skip_connection_type not in ("conv", "sum", "none"):
            raise ValueError("Unsupported skip connection type.")
        self.skip_connection_type = skip_connection_type
        self.relu_first = relu_first
        self.low_feat = low_feat
        c = channel_list
        if skip_connection_type == "conv":
            self.conv = nn.Conv2d(c[0], c[3], 1, stride=stride, bias=False)
            self.bn = norm_layer(c[3])
        self.sep_conv1 = SeparableConv2d(c[0], c[1], dilation=dilation, relu_first=relu_first,
                                         norm_layer=norm_layer)
        self.sep_conv2 = SeparableConv2d(c[1], c[2], dilation=dilation, relu_first=relu_first,
                                         norm_layer=norm_layer)
        self.sep_conv3 = SeparableConv2d(c[2], c[3], dilation=dilation, relu_first=relu_first,
                                         stride=stride, norm_layer=norm_layer)
        self.last_inp_channels = c[3]

    def forward(self, inputs):
        # "sum" blocks: `inputs` (a plain tensor) feeds the residual sum and sep_conv1 — the
        # identity path's gradient is handed to sep_conv1's depthwise backward, which adds it in
        # its store path (functional.GradFork) instead of autograd's element-wise add
        fork = None
        if (self.skip_connection_type == "sum" and self.relu_first and inputs.bn is None
                and not inputs.relu and torch.is_grad_enabled() and inputs.t.requires_grad):
            fork = F.GradFork()
        skip_in = inputs
        if self.skip_connection_type == "conv":
            # two consumers (first separable conv, shortcut conv): one 2-ary gradient sum on the
            # HIP kernel instead of autograd's `add` (functional.fork)
            t = F.fork(inputs.t, 2)
            inputs, skip_in = F.Act(t[0], inputs.bn, inputs.relu), F.Act(t[1], inputs.bn, inputs.relu)
        sc1 = self.sep_conv1(inputs, fork=fork) if fork is not None else self.sep_conv1(inputs)
        sc2 = self.sep_conv2(sc1)
        low = sc2
        if self.low_feat:  # the low-level feature leaves the block AND feeds sep_conv3
            t = F.fork(sc2.t, 2)
            sc2, low = F.Act(t[0], sc2.bn, sc2.relu), F.Act(t[1], sc2.bn, sc2.relu)
        residual = self.sep_conv3(sc2)
        if self.skip_connection_type == "conv":
            shortcut = F.conv_bn(skip_in, self.conv, self.bn)
            outputs = F.Act(F.materialize(residual, residual=shortcut))
        elif self.skip_connection_type == "sum":
            outputs = F.Act(F.materialize(residual, residual=inputs, fork=fork))
        else:
            outputs = residual
        return (outputs, low) if self.low_feat else outputs


class Xception65(nn.Module):
    """Entry flow (2 convs + 3 strided blocks), 16 middle blocks, exit flow (xception.py:54-165).
    Takes the NCHW float image, returns deferred NHWC activations (c1, c2, c3, c4)."""

    def __init__(self, norm_layer=nn.BatchNorm2d):
        super().__init__()
        os_ = cfg.MODEL.OUTPUT_STRIDE
        if os_ == 32:
            b3_stride, mid_dil, exit_dil, exit_stride = 2, 1, (1, 1), 2
        elif os_ == 16:
            b3_stride, mid_dil, exit_dil, exit_stride = 2, 1, (1, 2), 1
        elif os_ == 8:
            b3_stride, mid_dil, exit_dil, exit_stride = 1, 2, (2, 4), 1
        else:
            raise NotImplementedError
        self.conv1 = nn.Conv2d(3, 32, 3, stride=2, padding=1, bias=False)
        self.bn1 = norm_layer(32)
        self.relu = nn.ReLU()
        self.conv2 = nn.Conv2d(32, 64, 3, stride=1, padding=1, bias=False)
        self.bn2 = norm_layer(64)
        self.block1 = XceptionBlock([64, 128, 128, 128], stride=2, norm_layer=norm_layer)
        self.block2 = XceptionBlock([128, 256, 256, 256], stride=2, low_feat=True,
                                    norm_layer=norm_layer)
        self.block3 = XceptionBlock([256, 728, 728, 728], stride=b3_stride, low_feat=True,
                                    norm_layer=norm_layer)
        for i in range(4, 20):  # middle flow: block4 .. block19
            setattr(self, "block%d" % i,
                    XceptionBlock([728] * 4, dilation=mid_dil, skip_connection_type="sum",
                                  norm_layer=norm_layer))
        self.block20 = XceptionBlock([728, 728, 1024, 1024], stride=exit_stride,
                                     dilation=exit_dil[0], norm_layer=norm_layer)
        self.block21 = XceptionBlock([1024, 1536, 1536, 2048], dilation=exit_dil[1],
                                     skip_connection_type="none", relu_first=False,
                                     norm_layer=norm_layer)

    def forward(self, x):
        from ... import compute_dtype
        a = F.Act(F.image_to_nhwc(x, compute_dtype()))
        a = F.conv_bn(a, self.conv1, self.bn1)
        a.relu = True
        a = F.conv_bn(a, self.conv2, self.bn2)
        a.relu = True
        a = self.block1(a)
        a, c1 = self.block2(a)
        a, c2 = self.block3(a)
        for i in range(4, 20):
            a = getattr(self, "block%d" % i)(a)
        c3 = a
        a = self.block20(c3)
        c4 = self.block21(a)
        return c1, c2, c3, c4


@BACKBONE_REGISTRY.register()
def xception65(norm_layer=nn.BatchNorm2d):
    return Xception65(norm_layer=norm_layer)
