"""Import the reference implementation (read-only, /root/reference) in the dev container.

TEST INFRASTRUCTURE ONLY.  The reference needs ``torchvision`` and ``thop`` at import time
(segmentron/data/dataloader/seg_data_base.py:5, segmentron/models/pointrend.py:5,
segmentron/utils/visualize.py:8); neither is installed here, so in-memory stub modules are
registered first.  ``numpy.int`` (removed in numpy>=1.24) is used by
segmentron/models/backbones/hrnet.py:291.  Nothing is written to the reference tree.

The reference's ``cfg`` is a process-global singleton that freezes, so one model per process.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("SEGMENTRON_REFERENCE", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "segmentron"))


def _install_stubs():
    import numpy as np
    names = ("torchvision", "torchvision.models", "torchvision.models._utils",
             "torchvision.transforms")
    mods = {n: types.ModuleType(n) for n in names}
    mods["torchvision.models._utils"].IntermediateLayerGetter = type(
        "IntermediateLayerGetter", (), {})
    mods["torchvision"].models = mods["torchvision.models"]
    mods["torchvision.models"]._utils = mods["torchvision.models._utils"]
    mods["torchvision"].transforms = mods["torchvision.transforms"]
    for n, m in mods.items():
        sys.modules.setdefault(n, m)
    if "thop" not in sys.modules:
        thop = types.ModuleType("thop")
        thop.profile = lambda *a, **k: (0, 0)
        sys.modules["thop"] = thop
    if not hasattr(np, "int"):
        np.int = int


def build_reference_model(config_file, overrides=(), phase="test", seed=0):
    """Build one reference model from a reference yaml (path relative to the reference root).

    Mirrors what tools/train.py:209-223 / tools/eval.py:104-115 do before
    ``get_segmentation_model()`` (segmentron/models/model_zoo.py:17-24).
    """
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # make sure `segmentron` resolves to the reference, not to this repo's drop-in alias
    for name in [n for n in sys.modules if n == "segmentron" or n.startswith("segmentron.")]:
        mod = sys.modules[name]
        f = getattr(mod, "__file__", "") or ""
        if not f.startswith(REFERENCE_ROOT):
            del sys.modules[name]
    import torch
    from segmentron.config import cfg
    from segmentron.models.model_zoo import get_segmentation_model
    if "pspnet" in config_file:
        # F3: segmentron/models/pspnet.py:47 passes norm_kwargs into
        # _ConvBNReLU.__init__ (segmentron/modules/basic.py:66-67) which does not accept it.
        from segmentron.modules import basic
        orig = basic._ConvBNReLU.__init__

        def patched(self, *a, **k):
            k.pop("norm_kwargs", None)
            orig(self, *a, **k)
        basic._ConvBNReLU.__init__ = patched
    cfg.update_from_file(os.path.join(REFERENCE_ROOT, config_file))
    cfg.update_from_list(["TRAIN.BACKBONE_PRETRAINED", "False"] + list(overrides))
    cfg.PHASE = phase
    cfg.ROOT_PATH = REFERENCE_ROOT
    cfg.check_and_freeze()
    torch.manual_seed(seed)
    model = get_segmentation_model()
    return model, cfg


def apply_bn_attrs(model, cfg):
    """What segmentron/solver/optimizer.py:14-40 / tools/eval.py:50-53 do to BN eps/momentum."""
    import torch.nn as nn

    def _set(mods, attr, val):
        for _, m in mods:
            if isinstance(m, (nn.BatchNorm2d, nn.SyncBatchNorm)):
                setattr(m, attr, val)
    if cfg.MODEL.BN_EPS_FOR_ENCODER:
        _set(model.encoder.named_modules(), "eps", cfg.MODEL.BN_EPS_FOR_ENCODER)
    if cfg.MODEL.BN_EPS_FOR_DECODER:
        for name in model.decoder:
            _set(getattr(model, name).named_modules(), "eps", cfg.MODEL.BN_EPS_FOR_DECODER)
    if cfg.MODEL.BN_MOMENTUM and cfg.MODEL.BN_TYPE in ["BN"]:
        _set(model.named_modules(), "momentum", cfg.MODEL.BN_MOMENTUM)
