"""HRNet segmentation model — segmentron/models/hrnet_seg.py:16-60 (registered as 'HRNet';
forward returns a LIST with one [N,nclass,H,W] fp32 tensor, bilinear resizes use
align_corners=False)."""
import torch
import torch.nn as nn

from .. import functional as F
from ..config import cfg
from .model_zoo import MODEL_REGISTRY
from .segbase import SegBaseModel

__all__ = ["HighResolutionNet"]


@MODEL_REGISTRY.register(name="HRNet")
class HighResolutionNet(SegBaseModel):
    def __init__(self):
        super().__init__()
        self.hrnet_head = _HRNetHead(self.nclass, self.encoder.last_inp_channels)
        self.__setattr__("decoder", ["hrnet_head"])

    def forward(self, x):
        shape = x.shape[2:]
        feats = self.encoder(x)
        logits = self.hrnet_head(feats)
        out = F.logits_to_nchw(logits, shape, align_corners=False)
        return [out]


class _HRNetHead(nn.Module):
    """Bilinear-upsample the three coarse branches to branch 0, concat (each branch writes its
    channel slice of one buffer), 1x1 conv(+bias) -> BN -> ReLU -> KxK classifier(+bias)."""

    def __init__(self, nclass, last_inp_channels, norm_layer=nn.BatchNorm2d):
        super().__init__()
        k = cfg.MODEL.HRNET.FINAL_CONV_KERNEL
        self.last_layer = nn.Sequential(
            nn.Conv2d(last_inp_channels, last_inp_channels, 1, 1, 0),
            norm_layer(last_inp_channels),
            nn.ReLU(inplace=False),
            nn.Conv2d(last_inp_channels, nclass, k, 1, 1 if k == 3 else 0))
        self.nclass = nclass

    def forward(self, x):
        x0 = x[0]
        N, H, W, _ = x0.shape
        chans = [a.shape[-1] for a in x]
        t = x0.t
        buf = torch.empty((N, H, W, sum(chans)), dtype=t.dtype, device=t.device)
        parts = [F.materialize(x0, out=buf[..., :chans[0]], force=True)]
        o = chans[0]
        for a, c in zip(x[1:], chans[1:]):
            parts.append(F.bilinear(a, (H, W), align_corners=False, out=buf[..., o:o + c]))
            o += c
        a = F.conv_bn(F.Act(F.concat_alias(buf, parts)), self.last_layer[0], self.last_layer[1])
        a.relu = True
        vec = 8 if t.dtype == torch.bfloat16 else 4
        pitch = (self.nclass + 2 * vec - 1) // vec * vec
        out = torch.empty((N, H, W, pitch), dtype=t.dtype, device=t.device)[..., :self.nclass]
        return F.conv_bn(a, self.last_layer[3], None, out=out).t
