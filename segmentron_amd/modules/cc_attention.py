"""CrissCrossAttention — module tree / state_dict of segmentron/modules/cc_attention.py:50-72,
forward on the HIP criss-cross kernels (csrc/cca.hip) instead of the reference's CUDA extension
(`segmentron._C`, modules/csrc/criss_cross_attention)."""
import torch
import torch.nn as nn

from .. import functional as F

__all__ = ["CrissCrossAttention"]


class CrissCrossAttention(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.query_conv = nn.Conv2d(in_channels, in_channels // 8, 1)
        self.key_conv = nn.Conv2d(in_channels, in_channels // 8, 1)
        self.value_conv = nn.Conv2d(in_channels, in_channels, 1)
        self.gamma = nn.Parameter(torch.zeros(1))

    def forward(self, act):
        """act: deferred activation (its BN/ReLU is applied once: x feeds three projections and
        the residual).  Returns a plain NHWC tensor."""
        x = F.materialize(act)
        xa = F.Act(x)
        q = F.conv_bn(xa, self.query_conv).t
        k = F.conv_bn(xa, self.key_conv).t
        v = F.conv_bn(xa, self.value_conv).t
        return F.criss_cross_attention(q, k, v, x, self.gamma)
