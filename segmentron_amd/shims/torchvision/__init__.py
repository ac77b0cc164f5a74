"""Minimal stand-in for torchvision: exactly the names SegmenTron touches
(tools/train.py:17,37-40; tools/eval.py:17,33-36; tools/demo.py:9,33-36;
segmentron/data/dataloader/seg_data_base.py:5,45; segmentron/models/pointrend.py:5).
A real torchvision on sys.path takes precedence (this directory is appended last)."""
from . import models, transforms  # noqa: F401

__version__ = "0.0+segmentron_amd.shim"
