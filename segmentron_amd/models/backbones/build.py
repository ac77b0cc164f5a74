"""Backbone registry + factory — interface of segmentron/models/backbones/build.py:10-63.
Pretrained-backbone download is not reproduced (no network on the target boxes): a local
TRAIN.BACKBONE_PRETRAINED_PATH or the torch.hub checkpoint cache is used, anything else raises."""
import logging
import os

import torch

from ...config import cfg
from ...utils.registry import Registry

BACKBONE_REGISTRY = Registry("BACKBONE")
BACKBONE_REGISTRY.__doc__ = "Registry for backbones; objects are called as obj(norm_layer)."


# checkpoint file names the reference downloads into torch.hub's checkpoint dir (build.py:14-30)
_PRETRAINED_FILES = {
    "resnet18": "resnet18-5c106cde.pth", "resnet34": "resnet34-333f7ec4.pth",
    "resnet50": "resnet50-19c8e357.pth", "resnet101": "resnet101-5d3b4d8f.pth",
    "resnet152": "resnet152-b121ed2d.pth", "resnet50c": "resnet50-25c4b509.pth",
    "resnet101c": "resnet101-2a57e44d.pth", "resnet152c": "resnet152-0d43d698.pth",
    "xception65": "tf-xception65-270e81cf.pth",
    "hrnet_w18_small_v1": "hrnet-w18-small-v1-08f8ae64.pth",
    "mobilenet_v2": "mobilenetV2-15498621.pth",
}


def load_backbone_pretrained(model, backbone):
    """build.py:33-54: in the train phase with TRAIN.BACKBONE_PRETRAINED (the default) the
    reference loads ImageNet weights — from TRAIN.BACKBONE_PRETRAINED_PATH if that file exists,
    else from the URL / torch.hub checkpoint cache.  There is no network on the target boxes, so
    the cache is the only other source, and a request that cannot be served is an ERROR
    (silently training from random init would cost tens of mIoU points): opt out explicitly with
    `TRAIN.BACKBONE_PRETRAINED False`."""
    if not (cfg.PHASE == "train" and cfg.TRAIN.BACKBONE_PRETRAINED
            and not cfg.TRAIN.PRETRAINED_MODEL_PATH):
        return
    path = cfg.TRAIN.BACKBONE_PRETRAINED_PATH
    if path:
        if not os.path.isfile(path):
            raise FileNotFoundError("TRAIN.BACKBONE_PRETRAINED_PATH %r does not exist" % (path,))
    elif backbone not in _PRETRAINED_FILES:
        logging.info("{} has no pretrained model".format(backbone))  # build.py:41-43
        return
    else:
        path = os.path.join(torch.hub.get_dir(), "checkpoints", _PRETRAINED_FILES[backbone])
        if not os.path.isfile(path):
            raise RuntimeError(
                "backbone %s: TRAIN.BACKBONE_PRETRAINED is True, TRAIN.BACKBONE_PRETRAINED_PATH "
                "is empty, %s is not cached and URL download is unavailable offline.  Provide "
                "the file, or set TRAIN.BACKBONE_PRETRAINED False to train from random init."
                % (backbone, path))
    logging.info("Load backbone pretrained model from {}".format(path))
    logging.info(model.load_state_dict(torch.load(path, map_location="cpu"), strict=False))


def get_segmentation_backbone(backbone, norm_layer=torch.nn.BatchNorm2d):
    model = BACKBONE_REGISTRY.get(backbone)(norm_layer)
    load_backbone_pretrained(model, backbone)
    return model
