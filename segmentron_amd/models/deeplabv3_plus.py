"""DeepLabv3+ — module tree / state_dict of segmentron/models/deeplabv3_plus.py:13-75."""
import torch
import torch.nn as nn

from .. import functional as F
from ..config import cfg
from ..modules import SeparableConv2d, _ASPP, _ConvBNReLU, _FCNHead
from .model_zoo import MODEL_REGISTRY
from .segbase import SegBaseModel

__all__ = ["DeepLabV3Plus"]


@MODEL_REGISTRY.register(name="DeepLabV3_Plus")
class DeepLabV3Plus(SegBaseModel):
    """encoder -> ASPP(c4) -> x-up to c1 -> cat(48-ch 1x1(c1)) -> 2 x SepConv -> 1x1 classifier
    -> x-up to the input size.  forward() takes the NCHW float image and returns a tuple whose
    [0] is the [N, nclass, H, W] float32 logits (tools/train.py:188, segbase.py:69)."""

    def __init__(self):
        super().__init__()
        if self.backbone.startswith("mobilenet"):
            c1_channels, c4_channels = 24, 320
        else:
            c1_channels, c4_channels = 256, 2048
        self.head = _DeepLabHead(self.nclass, c1_channels=c1_channels, c4_channels=c4_channels)
        if self.aux:
            self.auxlayer = _FCNHead(728, self.nclass)
        self.__setattr__("decoder", ["head", "auxlayer"] if self.aux else ["head"])

    def forward(self, x):
        size = x.shape[2:]
        lazy = F.want_lazy_logits(self.training)  # see functional.LogitsView
        c1, _, c3, c4 = self.encoder(x)
        y = self.head(c4, c1)  # NHWC logits at c1 resolution
        outputs = [F.logits_to_nchw(y, size, align_corners=True, lazy=lazy)]
        if self.aux:
            outputs.append(F.logits_to_nchw(self.auxlayer(c3), size, align_corners=True, lazy=lazy))
        return tuple(outputs)


class _DeepLabHead(nn.Module):
    def __init__(self, nclass, c1_channels=256, c4_channels=2048, norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.use_aspp = cfg.MODEL.DEEPLABV3_PLUS.USE_ASPP
        self.use_decoder = cfg.MODEL.DEEPLABV3_PLUS.ENABLE_DECODER
        last_channels = c4_channels
        if self.use_aspp:
            self.aspp = _ASPP(c4_channels, 256)
            last_channels = 256
        if self.use_decoder:
            self.c1_block = _ConvBNReLU(c1_channels, 48, 1, norm_layer=norm_layer)
            last_channels += 48
        self.block = nn.Sequential(
            SeparableConv2d(last_channels, 256, 3, norm_layer=norm_layer, relu_first=False),
            SeparableConv2d(256, 256, 3, norm_layer=norm_layer, relu_first=False),
            nn.Conv2d(256, nclass, 1))
        self.nclass = nclass

    def forward(self, x, c1):
        mul = None
        if self.use_aspp:
            x, mul = self.aspp(x)
        if self.use_decoder:
            N, H1, W1, _ = c1.shape
            cx = x.shape[-1]
            buf = torch.empty((N, H1, W1, cx + 48), dtype=x.t.dtype, device=x.t.device)
            up = F.bilinear(x, (H1, W1), chan_mul=mul, out=buf[..., :cx])
            low = F.materialize(self.c1_block(c1), out=buf[..., cx:cx + 48])
            x = F.Act(F.concat_alias(buf, [up, low]))
        elif mul is not None:
            x = F.Act(F.materialize(x, chan_mul=mul))
        x = self.block[1](self.block[0](x))
        N, H, W, _ = x.shape
        vec = 8 if x.t.dtype == torch.bfloat16 else 4
        pitch = (self.nclass + 2 * vec - 1) // vec * vec
        out = torch.empty((N, H, W, pitch), dtype=x.t.dtype, device=x.t.device)[..., :self.nclass]
        return F.conv_bn(x, self.block[2], None, out=out).t
