"""DANet — module tree / state_dict of segmentron/models/danet.py:14-89 ("Dual Attention Network
for Scene Segmentation"), forward on the HIP path: ResNet encoder (multi-grid layer4 when
cfg.MODEL.DANET.MULTI_GRID), position- and channel-attention heads (modules/module.py PAM_Module /
CAM_Module: MFMA GEMMs + row softmax), three classifiers — fused, position, channel — each
bilinearly upsampled to the input size."""
import torch
import torch.nn as nn

from .. import functional as F
from ..modules import CAM_Module, PAM_Module, _FCNHead
from .model_zoo import MODEL_REGISTRY
from .segbase import SegBaseModel

__all__ = ["DANet"]


@MODEL_REGISTRY.register()
class DANet(SegBaseModel):
    def __init__(self):
        super().__init__()
        self.head = DANetHead(2048, self.nclass)
        if self.aux:
            self.auxlayer = _FCNHead(728, self.nclass)  # (as the reference, danet.py:23; unused)
        self.__setattr__("decoder", ["head", "auxlayer"] if self.aux else ["head"])

    def forward(self, x):
        size = x.shape[2:]
        lazy = F.want_lazy_logits(self.training)
        _, _, _, c4 = self.encoder(x)
        outs = self.head(c4)
        return tuple(F.logits_to_nchw(y, size, align_corners=True, lazy=lazy) for y in outs)


def _cbr(seq):
    return lambda act: _relu(F.conv_bn(act, seq[0], seq[1]))


def _relu(a):
    a.relu = True
    return a


class DANetHead(nn.Module):
    def __init__(self, in_channels, out_channels, norm_layer=nn.BatchNorm2d):
        super().__init__()
        inter = in_channels // 4

        def block(cin):
            return nn.Sequential(nn.Conv2d(cin, inter, 3, padding=1, bias=False),
                                 norm_layer(inter), nn.ReLU())
        self.conv5a = block(in_channels)
        self.conv5c = block(in_channels)
        self.sa = PAM_Module(inter)
        self.sc = CAM_Module(inter)
        self.conv51 = block(inter)
        self.conv52 = block(inter)
        self.conv6 = nn.Sequential(nn.Dropout2d(0.1, False), nn.Conv2d(512, out_channels, 1))
        self.conv7 = nn.Sequential(nn.Dropout2d(0.1, False), nn.Conv2d(512, out_channels, 1))
        self.conv8 = nn.Sequential(nn.Dropout2d(0.1, False), nn.Conv2d(512, out_channels, 1))
        self.nclass = out_channels

    def _classify(self, t, seq):
        """Dropout2d + 1x1 (+bias) on a plain NHWC tensor -> channel-padded NHWC logits."""
        N, H, W, C = t.shape
        p = seq[0].p
        if self.training and p > 0.0:
            mul = (torch.rand((N, C), device=t.device) >= p).float() / (1.0 - p)
            t = F.materialize(F.Act(t), chan_mul=mul)
        vec = 8 if t.dtype == torch.bfloat16 else 4
        pitch = (self.nclass + 2 * vec - 1) // vec * vec
        out = torch.empty((N, H, W, pitch), dtype=t.dtype, device=t.device)[..., :self.nclass]
        return F.conv_bn(F.Act(t), seq[1], None, out=out).t

    def forward(self, c4):
        x = F.Act(F.materialize(c4))  # shared by conv5a / conv5c
        sa_conv = _cbr(self.conv51)(F.Act(self.sa(F.materialize(_cbr(self.conv5a)(x)))))
        sc_conv = _cbr(self.conv52)(F.Act(self.sc(F.materialize(_cbr(self.conv5c)(x)))))
        sa_t, sc_t = F.materialize(sa_conv), F.materialize(sc_conv)
        feat_sum = F.materialize(F.Act(sa_t), residual=F.Act(sc_t), force=True)
        return (self._classify(feat_sum, self.conv8), self._classify(sa_t, self.conv6),
                self._classify(sc_t, self.conv7))
