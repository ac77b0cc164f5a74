"""CCNet — module tree / state_dict of segmentron/models/ccnet.py:11-86 (disabled in the
reference, models/__init__.py:11, because its CUDA extension is not built by default), forward on
the HIP path: ResNet encoder, recurrent criss-cross attention head (csrc/cca.hip), FCN aux head."""
import torch
import torch.nn as nn

from .. import functional as F
from ..config import cfg
from ..modules import _FCNHead
from ..modules.cc_attention import CrissCrossAttention
from .model_zoo import MODEL_REGISTRY
from .segbase import SegBaseModel

__all__ = ["CCNet"]


@MODEL_REGISTRY.register()
class CCNet(SegBaseModel):
    def __init__(self):
        super().__init__()
        self.head = _CCHead(self.nclass, norm_layer=self.norm_layer)
        if self.aux:
            self.auxlayer = _FCNHead(1024, self.nclass, norm_layer=self.norm_layer)
        self.__setattr__("decoder", ["head", "auxlayer"] if self.aux else ["head"])

    def forward(self, x):
        size = tuple(x.shape[2:])
        _, _, c3, c4 = self.base_forward(x)
        lazy = F.want_lazy_logits(self.training)  # see functional.LogitsView
        outputs = [F.logits_to_nchw(self.head(c4), size, lazy=lazy)]
        if self.aux:
            outputs.append(F.logits_to_nchw(self.auxlayer(c3), size, lazy=lazy))
        return tuple(outputs)


class _CCHead(nn.Module):
    def __init__(self, nclass, norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.rcca = _RCCAModule(2048, 512, norm_layer)
        self.out = nn.Conv2d(512, nclass, 1)
        self.nclass = nclass

    def forward(self, act):
        a, mul = self.rcca(act)
        # bottleneck BN (no ReLU) + the Dropout2d channel mask are applied once, then the biased
        # 1x1 classifier writes the channel-padded logits buffer
        t = F.materialize(a, chan_mul=mul)
        N, H, W, _ = t.shape
        vec = 8 if t.dtype == torch.bfloat16 else 4
        pitch = (self.nclass + 2 * vec - 1) // vec * vec
        out = torch.empty((N, H, W, pitch), dtype=t.dtype, device=t.device)[..., :self.nclass]
        return F.conv_bn(F.Act(t), self.out, None, out=out).t


class _RCCAModule(nn.Module):
    """conva -> `recurrence` x criss-cross attention (shared weights) -> convb -> cat with the
    input -> bottleneck (ccnet.py:56-86)."""

    def __init__(self, in_channels, out_channels, norm_layer):
        super().__init__()
        self.recurrence = cfg.MODEL.CCNET.RECURRENCE
        inter = in_channels // 4
        self.conva = nn.Sequential(nn.Conv2d(in_channels, inter, 3, padding=1, bias=False),
                                   norm_layer(inter), nn.ReLU(True))
        self.cca = CrissCrossAttention(inter)
        self.convb = nn.Sequential(nn.Conv2d(inter, inter, 3, padding=1, bias=False),
                                   norm_layer(inter), nn.ReLU(True))
        self.bottleneck = nn.Sequential(
            nn.Conv2d(in_channels + inter, out_channels, 3, padding=1, bias=False),
            norm_layer(out_channels), nn.Dropout2d(0.1))
        self.in_channels, self.inter = in_channels, inter

    def forward(self, act):
        """-> (Act with the bottleneck BN pending, Dropout2d multiplier [N, C] | None)."""
        x = F.materialize(act)
        N, H, W, C = x.shape
        a = F.conv_bn(F.Act(x), self.conva[0], self.conva[1])
        a.relu = True
        for _ in range(self.recurrence):
            a = F.Act(self.cca(a))
        b = F.conv_bn(a, self.convb[0], self.convb[1])
        b.relu = True
        # torch.cat([x, out], 1): both producers write their channel slice of one buffer
        buf = torch.empty((N, H, W, C + self.inter), dtype=x.dtype, device=x.device)
        parts = [F.materialize(F.Act(x), out=buf[..., :C], force=True),
                 F.materialize(b, out=buf[..., C:])]
        y = F.conv_bn(F.Act(F.concat_alias(buf, parts)), self.bottleneck[0], self.bottleneck[1])
        mul = None
        p = self.bottleneck[2].p
        if self.training and p > 0.0:
            keep = torch.rand((N, y.shape[-1]), device=x.device) >= p
            mul = keep.float() / (1.0 - p)
        return y, mul
