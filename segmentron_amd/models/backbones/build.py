"""Backbone registry + factory — interface of segmentron/models/backbones/build.py:10-63.
Pretrained-backbone download is not reproduced (no network on the target boxes); a local
TRAIN.BACKBONE_PRETRAINED_PATH is honoured exactly as build.py:35-40 does."""
import logging
import os

import torch

from ...config import cfg
from ...utils.registry import Registry

BACKBONE_REGISTRY = Registry("BACKBONE")
BACKBONE_REGISTRY.__doc__ = "Registry for backbones; objects are called as obj(norm_layer)."


def load_backbone_pretrained(model, backbone):
    if cfg.PHASE == "train" and cfg.TRAIN.BACKBONE_PRETRAINED \
            and not cfg.TRAIN.PRETRAINED_MODEL_PATH:
        path = cfg.TRAIN.BACKBONE_PRETRAINED_PATH
        if path and os.path.isfile(path):
            logging.info("Load backbone pretrained model from {}".format(path))
            logging.info(model.load_state_dict(torch.load(path, map_location="cpu"),
                                               strict=False))
        else:
            logging.warning("backbone %s: TRAIN.BACKBONE_PRETRAINED is set but no local "
                            "TRAIN.BACKBONE_PRETRAINED_PATH file exists and URL download is "
                            "unavailable offline — using random init", backbone)


def get_segmentation_backbone(backbone, norm_layer=torch.nn.BatchNorm2d):
    model = BACKBONE_REGISTRY.get(backbone)(norm_layer)
    load_backbone_pretrained(model, backbone)
    return model
