"""Composite blocks — interface of segmentron/modules/module.py:32-77 (_ASPP)."""
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import functional as F
from ..config import cfg
from .basic import SeparableConv2d, _ConvBNReLU

__all__ = ["_ASPP", "_FCNHead", "PyramidPooling", "PAM_Module", "CAM_Module"]


class _FCNHead(nn.Module):
    """3x3 conv C -> C/4 + BN + ReLU + Dropout(0.1) + 1x1 -> nclass (module.py:13-26).
    Returns NHWC logits (a view of a channel-padded buffer)."""

    def __init__(self, in_channels, channels, norm_layer=nn.BatchNorm2d):
        super().__init__()
        inter = in_channels // 4
        # r06: hidden widths that are not a multiple of the 16-byte channel vector — DeepLabv3+ on
        # xception with SOLVER.AUX True builds _FCNHead(728, nclass): 182 channels
        # (deeplabv3_plus.py:29-30) — are PADDED inside the module (184): the extra output channels
        # of the 3x3 conv have zero weights, their BatchNorm maps 0 to beta = 0, ReLU keeps 0, and
        # the classifier's extra input columns are zero, so they carry neither signal nor gradient
        # and stay at their initial values under SGD.  state_dict() / load_state_dict() speak the
        # reference's shapes (hooks below): same keys, same tensors, checkpoints interchange.
        vec = 8
        inter_p = (inter + vec - 1) // vec * vec
        self.inter, self.inter_p = inter, inter_p
        self.block = nn.Sequential(
            nn.Conv2d(in_channels, inter_p, 3, padding=1, bias=False),
            norm_layer(inter_p),
            nn.ReLU(inplace=True),
            nn.Dropout(0.1),
            nn.Conv2d(inter_p, channels, 1))
        self.channels = channels
        if inter_p != inter:
            with torch.no_grad():
                self.block[0].weight[inter:].zero_()
                self.block[4].weight[:, inter:].zero_()
            self._register_state_dict_hook(_FCNHead._slice_state)
            self._register_load_state_dict_pre_hook(self._pad_state)

    # (key suffix, dimension that carries the hidden channels, value of the padding)
    _PADDED = (("block.0.weight", 0, 0.0), ("block.1.weight", 0, 1.0), ("block.1.bias", 0, 0.0),
               ("block.1.running_mean", 0, 0.0), ("block.1.running_var", 0, 1.0),
               ("block.4.weight", 1, 0.0))

    @staticmethod
    def _slice_state(module, state_dict, prefix, local_metadata):
        for suffix, dim, _ in _FCNHead._PADDED:
            k = prefix + suffix
            if k in state_dict and state_dict[k].shape[dim] == module.inter_p:
                state_dict[k] = state_dict[k].narrow(dim, 0, module.inter)
        return state_dict

    def _pad_state(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                   error_msgs):
        for suffix, dim, fill in _FCNHead._PADDED:
            k = prefix + suffix
            if k in state_dict and state_dict[k].shape[dim] == self.inter:
                t = state_dict[k]
                shape = list(t.shape)
                shape[dim] = self.inter_p - self.inter
                state_dict[k] = torch.cat([t, torch.full(shape, fill, dtype=t.dtype, device=t.device)], dim)

    def forward(self, act):
        return head_tail(act, self.block[0], self.block[1], self.block[3], self.block[4],
                         self.training, self.channels)


def head_tail(act, conv, bn, dropout, classifier, training, nclass):
    """conv -> BN -> ReLU -> Dropout -> 1x1 classifier (+bias), shared by _FCNHead / _PSPHead."""
    a = F.conv_bn(act, conv, bn)
    a.relu = True
    p = dropout.p
    if training and p > 0.0:
        mask = F.dropout_mask(a.t.shape, p, a.t.dtype, a.t.device)
        a = F.Act(F.materialize(a, elem_mul=mask))
    N, H, W, _ = a.shape
    vec = 8 if a.t.dtype == torch.bfloat16 else 4
    pitch = (nclass + 2 * vec - 1) // vec * vec
    out = torch.empty((N, H, W, pitch), dtype=a.t.dtype, device=a.t.device)[..., :nclass]
    return F.conv_bn(a, classifier, None, out=out).t


class PyramidPooling(nn.Module):
    """Adaptive pools (1,2,3,6) -> 1x1 C -> C/4 + BN + ReLU -> bilinear up -> concat with x
    (module.py:82-97).  Every branch writes its slice of one [N,H,W,2C] buffer."""

    def __init__(self, in_channels, sizes=(1, 2, 3, 6), norm_layer=nn.BatchNorm2d, **kwargs):
        super().__init__()
        out_channels = int(in_channels / 4)
        self.sizes = tuple(sizes)
        self.avgpools = nn.ModuleList([nn.AdaptiveAvgPool2d(s) for s in sizes])
        self.convs = nn.ModuleList([_ConvBNReLU(in_channels, out_channels, 1,
                                                norm_layer=norm_layer) for _ in sizes])
        self.in_channels, self.out_channels = in_channels, out_channels

    def forward(self, act):
        x = F.materialize(act)
        N, H, W, C = x.shape
        oc = self.out_channels
        buf = torch.empty((N, H, W, C + oc * len(self.sizes)), dtype=x.dtype, device=x.device)
        parts = [F.materialize(F.Act(x), out=buf[..., :C], force=True)]
        for i, (size, conv) in enumerate(zip(self.sizes, self.convs)):
            # (bins, 1x1 convolution and its few-sample BatchNorm in float32 whatever the compute
            # dtype, rounded once behind BN + ReLU: functional.global_avg_pool)
            a = conv(F.Act(F.adaptive_avg_pool(x, size, keep_fp32=True)))
            if x.dtype != torch.float32:
                a = F.Act(F.materialize(a).to(x.dtype))
            parts.append(F.bilinear(a, (H, W), out=buf[..., C + i * oc:C + (i + 1) * oc]))
        return F.Act(F.concat_alias(buf, parts))


class _ASPP(nn.Module):
    """Atrous spatial pyramid pooling: image-pooling branch, 1x1 branch, three dilated separable
    branches -> concat(1280) -> 1x1 -> BN -> ReLU -> Dropout2d (module.py:32-77).

    MI355X form: c4 is materialised once (its BN+ReLU is shared by five consumers), every branch
    writes straight into its channel slice of one [N,H,W,1280] buffer (no torch.cat copy), the
    projection's BN+ReLU and the Dropout2d channel mask are deferred to the consumer (the
    decoder's bilinear upsample)."""

    def __init__(self, in_channels=2048, out_channels=256):
        super().__init__()
        output_stride = cfg.MODEL.OUTPUT_STRIDE
        if output_stride in (16, 32):
            dilations = [6, 12, 18]
        elif output_stride == 8:
            dilations = [12, 24, 36]
        else:
            raise NotImplementedError
        oc = out_channels
        self.aspp0 = nn.Sequential(OrderedDict([
            ("conv", nn.Conv2d(in_channels, oc, 1, bias=False)), ("bn", nn.BatchNorm2d(oc)),
            ("relu", nn.ReLU(inplace=True))]))
        self.aspp1 = SeparableConv2d(in_channels, oc, dilation=dilations[0], relu_first=False)
        self.aspp2 = SeparableConv2d(in_channels, oc, dilation=dilations[1], relu_first=False)
        self.aspp3 = SeparableConv2d(in_channels, oc, dilation=dilations[2], relu_first=False)
        self.image_pooling = nn.Sequential(OrderedDict([
            ("gap", nn.AdaptiveAvgPool2d((1, 1))),
            ("conv", nn.Conv2d(in_channels, oc, 1, bias=False)), ("bn", nn.BatchNorm2d(oc)),
            ("relu", nn.ReLU(inplace=True))]))
        self.conv = nn.Conv2d(oc * 5, oc, 1, bias=False)
        self.bn = nn.BatchNorm2d(oc)
        self.relu = nn.ReLU(inplace=True)
        self.dropout = nn.Dropout2d(p=0.1)
        self.out_channels = oc

    def forward(self, c4):
        """c4: Act.  Returns (Act [N,H,W,256] with BN+ReLU pending, Dropout2d multiplier|None)."""
        x = F.materialize(c4)
        N, H, W, _ = x.shape
        oc = self.out_channels
        # five consumers of one tensor: their gradients meet in ONE n-ary sum (functional.fork)
        xs = F.fork(x, 5)
        buf = torch.empty((N, H, W, 5 * oc), dtype=x.dtype, device=x.device)
        # image pooling: gap -> 1x1 -> BN (statistics over the batch) -> ReLU -> broadcast
        # (in float32 whatever the compute dtype — functional.global_avg_pool — and rounded ONCE,
        # after BatchNorm + ReLU, where nothing cancels any more)
        pooled = F.conv_bn(F.Act(F.global_avg_pool(xs[0], keep_fp32=True)),
                           self.image_pooling.conv, self.image_pooling.bn)
        pooled.relu = True
        if x.dtype != torch.float32:
            pooled = F.Act(F.materialize(pooled).to(x.dtype))
        parts = [F.bilinear(pooled, (H, W), out=buf[..., 0:oc])]
        b0 = F.conv_bn(F.Act(xs[1]), self.aspp0.conv, self.aspp0.bn)
        b0.relu = True
        parts.append(F.materialize(b0, out=buf[..., oc:2 * oc]))
        for i, branch in enumerate((self.aspp1, self.aspp2, self.aspp3)):
            parts.append(F.materialize(branch(F.Act(xs[2 + i])),
                                       out=buf[..., (2 + i) * oc:(3 + i) * oc]))
        cat = F.concat_alias(buf, parts)
        y = F.conv_bn(F.Act(cat), self.conv, self.bn)
        y.relu = True
        mul = None
        p = self.dropout.p
        if self.training and p > 0.0:
            keep = torch.rand((N, oc), device=x.device) >= p
            mul = keep.float() / (1.0 - p)
        return y, mul


class PAM_Module(nn.Module):
    """Position attention module (module.py:100-130): out = gamma * (V softmax(Q^T K)^T) + x.
    Parameter container with the reference's names; forward on functional.position_attention
    (MFMA GEMMs + the row-softmax kernel).  Takes / returns plain NHWC tensors."""

    def __init__(self, in_dim):
        super().__init__()
        self.chanel_in = in_dim
        self.query_conv = nn.Conv2d(in_channels=in_dim, out_channels=in_dim // 8, kernel_size=1)
        self.key_conv = nn.Conv2d(in_channels=in_dim, out_channels=in_dim // 8, kernel_size=1)
        self.value_conv = nn.Conv2d(in_channels=in_dim, out_channels=in_dim, kernel_size=1)
        self.gamma = nn.Parameter(torch.zeros(1))
        self.softmax = nn.Softmax(dim=-1)

    def forward(self, x):
        xa = F.Act(x)
        q = F.conv_bn(xa, self.query_conv).t
        k = F.conv_bn(xa, self.key_conv).t
        v = F.conv_bn(xa, self.value_conv).t
        return F.position_attention(q, k, v, x, self.gamma)


class CAM_Module(nn.Module):
    """Channel attention module (module.py:133-162): out = gamma * (softmax(max - X X^T) X) + x."""

    def __init__(self, in_dim):
        super().__init__()
        self.chanel_in = in_dim
        self.gamma = nn.Parameter(torch.zeros(1))
        self.softmax = nn.Softmax(dim=-1)

    def forward(self, x):
        return F.channel_attention(x, self.gamma)
