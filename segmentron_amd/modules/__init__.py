"""Seg NN modules (names as in segmentron/modules/__init__.py)."""
from .basic import *  # noqa: F401,F403
from .module import *  # noqa: F401,F403
from .batch_norm import get_norm  # noqa: F401
