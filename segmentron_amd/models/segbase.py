"""SegBaseModel — contract of segmentron/models/segbase.py:16-79."""
import functools
import math

import torch
import torch.nn as nn
import torch.nn.functional as TF

from .. import functional as F
from ..config import cfg
from ..data.dataloader import datasets
from ..modules import get_norm
from .backbones import get_segmentation_backbone

__all__ = ["SegBaseModel"]


class SegBaseModel(nn.Module):
    def __init_subclass__(cls, **kw):
        """Every model's own `forward` runs inside a functional.bn_counter_scope: the
        `num_batches_tracked` increments of its BatchNorms (torch nn.BatchNorm2d semantics, e.g.
        reference ccnet.py:57-86) are applied as one launch when the forward returns and dropped
        when it raises — whether it is entered through `model(x)`, `evaluate()`'s
        `self.forward(x)` or a graph capture.  No model file carries the flush itself."""
        super().__init_subclass__(**kw)
        fwd = cls.__dict__.get("forward")
        if fwd is not None and not getattr(fwd, "_bn_counter_scoped", False):
            @functools.wraps(fwd)
            def forward(self, *args, **kwargs):
                with F.bn_counter_scope():
                    return fwd(self, *args, **kwargs)
            forward._bn_counter_scoped = True
            cls.forward = forward

    def __init__(self, need_backbone=True):
        super().__init__()
        self.nclass = datasets[cfg.DATASET.NAME].NUM_CLASS
        self.aux = cfg.SOLVER.AUX
        self.norm_layer = get_norm(cfg.MODEL.BN_TYPE)
        self.backbone = None
        self.encoder = None
        if need_backbone:
            self.get_backbone()

    def get_backbone(self):
        self.backbone = cfg.MODEL.BACKBONE.lower()
        self.encoder = get_segmentation_backbone(self.backbone, self.norm_layer)

    def base_forward(self, x):
        return self.encoder(x)

    def demo(self, x):
        pred = self.forward(x)
        return pred[0] if self.aux else pred

    def evaluate(self, image):
        """Multi-scale / flip / pad / crop inference (segbase.py:44-79).  The image-space glue
        (resize of the 3-channel image, zero pad, crop, flip of NCHW score maps) is host
        plumbing on torch tensors; every forward pass runs on the HIP kernels."""
        scales, flip = cfg.TEST.SCALES, cfg.TEST.FLIP
        crop = cfg.TEST.CROP_SIZE
        if crop is not None and not isinstance(crop, (list, tuple)):
            crop = (crop, crop)
        _, _, h, w = image.shape
        base = max(h, w)
        scores = None
        for scale in scales:
            long_size = int(math.ceil(base * scale))
            if h > w:
                height, width = long_size, int(1.0 * w * long_size / h + 0.5)
            else:
                width, height = long_size, int(1.0 * h * long_size / w + 0.5)
            cur = _resize(image, height, width)
            if crop is not None:
                assert crop[0] >= h and crop[1] >= w
                ch, cw = int(math.ceil(crop[0] * scale)), int(math.ceil(crop[1] * scale))
                padh, padw = max(ch - height, 0), max(cw - width, 0)
                # as the reference's _pad_image (segbase.py:88-95): F.pad(img, (0, padh, 0, padw))
                # — i.e. the WIDTH grows by padh and the HEIGHT by padw (kept for parity; equal
                # for the configs' square-ish crops, e.g. 1025x2049 around 1024x2048)
                if padh or padw:
                    cur = TF.pad(cur, (0, padh, 0, padw))
            out = _crop(self.forward(cur)[0], height, width)
            if flip:
                out = out + _crop(self.forward(cur.flip(3))[0].flip(3), height, width)
            score = _resize(out, h, w)
            scores = score if scores is None else scores + score
        return scores


def _crop(out, h, w):
    """out[..., :h, :w]; the identity keeps a pending LogitsView pending."""
    if out.shape[2] == h and out.shape[3] == w:
        return out
    return out[..., :h, :w]


def _resize(img, h, w):
    if img.shape[2] == h and img.shape[3] == w:
        return img
    return TF.interpolate(img, size=[h, w], mode="bilinear", align_corners=True)
