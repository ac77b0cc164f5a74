"""PSPNet — segmentron/models/pspnet.py:13-58.  (The reference's `_PSPHead` passes a stray
`norm_kwargs` into `_ConvBNReLU` and cannot be constructed at HEAD — SURVEY.md F3; the module tree
here is the one that constructor intends, with the same state_dict keys.)"""
import torch
import torch.nn as nn

from .. import functional as F
from ..modules import PyramidPooling, _FCNHead
from ..modules.module import head_tail
from .model_zoo import MODEL_REGISTRY
from .segbase import SegBaseModel

__all__ = ["PSPNet"]


@MODEL_REGISTRY.register()
class PSPNet(SegBaseModel):
    def __init__(self):
        super().__init__()
        self.head = _PSPHead(self.nclass)
        if self.aux:
            self.auxlayer = _FCNHead(1024, self.nclass)
        self.__setattr__("decoder", ["head", "auxlayer"] if self.aux else ["head"])

    def forward(self, x):
        size = x.shape[2:]
        lazy = F.want_lazy_logits(self.training)  # see functional.LogitsView
        _, _, c3, c4 = self.encoder(x)
        outputs = [F.logits_to_nchw(self.head(c4), size, align_corners=True, lazy=lazy)]
        if self.aux:
            outputs.append(F.logits_to_nchw(self.auxlayer(c3), size, align_corners=True, lazy=lazy))
        return tuple(outputs)


class _PSPHead(nn.Module):
    def __init__(self, nclass, norm_layer=nn.BatchNorm2d, norm_kwargs=None, **kwargs):
        super().__init__()
        self.psp = PyramidPooling(2048, norm_layer=norm_layer)
        self.block = nn.Sequential(
            nn.Conv2d(4096, 512, 3, padding=1, bias=False),
            norm_layer(512, **({} if norm_kwargs is None else norm_kwargs)),
            nn.ReLU(True),
            nn.Dropout(0.1),
            nn.Conv2d(512, nclass, 1))
        self.nclass = nclass

    def forward(self, act):
        x = self.psp(act)
        return head_tail(x, self.block[0], self.block[1], self.block[3], self.block[4],
                         self.training, self.nclass)
