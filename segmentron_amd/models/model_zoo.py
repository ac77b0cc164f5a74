"""Model registry + factory — interface of segmentron/models/model_zoo.py:8-46."""
import logging
from collections import OrderedDict

import torch

from ..config import cfg
from ..utils.registry import Registry

MODEL_REGISTRY = Registry("MODEL")
MODEL_REGISTRY.__doc__ = "Registry for whole segmentation models; objects are called as obj()."


def get_segmentation_model():
    """Build the model named by cfg.MODEL.MODEL_NAME (case-sensitive, model_zoo.py:22)."""
    model = MODEL_REGISTRY.get(cfg.MODEL.MODEL_NAME)()
    load_model_pretrain(model)
    from .. import graph
    if graph.transparent_graph_requested():
        # SEGMENTRON_HIP_GRAPH=1: train-mode forward / backward replay captured HIP graphs behind
        # the unchanged tools/train.py loop (segmentron_amd/graph.py TransparentTrainGraph)
        graph.TransparentTrainGraph.install(model)
    return model


def load_model_pretrain(model):
    """Whole-model weights, strict=False (model_zoo.py:27-46): train phase loads
    TRAIN.PRETRAINED_MODEL_PATH keeping only shape-matching tensors, otherwise
    TEST.TEST_MODEL_PATH."""
    if cfg.PHASE == "train":
        path = cfg.TRAIN.PRETRAINED_MODEL_PATH
        if not path:
            return
        logging.info("load pretrained model from {}".format(path))
        own = model.state_dict()
        good, bad = OrderedDict(), []
        for k, v in torch.load(path, map_location="cpu").items():
            (good.__setitem__(k, v) if k in own and v.shape == own[k].shape else bad.append(k))
        logging.info("Shape unmatched weights: {}".format(bad))
        logging.info(model.load_state_dict(good, strict=False))
    elif cfg.TEST.TEST_MODEL_PATH:
        logging.info("load test model from {}".format(cfg.TEST.TEST_MODEL_PATH))
        logging.info(model.load_state_dict(
            torch.load(cfg.TEST.TEST_MODEL_PATH, map_location="cpu"), strict=False))
