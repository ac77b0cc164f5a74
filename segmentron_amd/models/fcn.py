"""FCN — segmentron/models/fcn.py:12-34 (head hard-codes 2048 input channels, i.e. needs a
Bottleneck ResNet; `resnet18` fails exactly as in the reference)."""
import torch

from .. import functional as F
from ..modules import _FCNHead
from .model_zoo import MODEL_REGISTRY
from .segbase import SegBaseModel

__all__ = ["FCN"]


@MODEL_REGISTRY.register()
class FCN(SegBaseModel):
    def __init__(self):
        super().__init__()
        self.head = _FCNHead(2048, self.nclass)
        if self.aux:
            self.auxlayer = _FCNHead(1024, self.nclass)
        self.__setattr__("decoder", ["head", "auxlayer"] if self.aux else ["head"])

    def forward(self, x):
        size = x.shape[2:]
        lazy = F.want_lazy_logits(self.training)  # see functional.LogitsView
        _, _, c3, c4 = self.base_forward(x)
        outputs = [F.logits_to_nchw(self.head(c4), size, align_corners=True, lazy=lazy)]
        if self.aux:
            outputs.append(F.logits_to_nchw(self.auxlayer(c3), size, align_corners=True, lazy=lazy))
        return tuple(outputs)
