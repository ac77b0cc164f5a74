"""segmentron_amd — MI355X-native (gfx950) implementation of SegmenTron's dense-convolution hot
path behind the reference's `segmentron.models` registry + `cfg` API.

The arithmetic lives in libsegmentron_hip.so (segmentron_amd/csrc, C-ABI in
include/segmentron_hip.h); this package is the host-side mirror of the reference interface.
There is no CPU / PyTorch-op fallback: building a model works anywhere, running it needs the HIP
library and a HIP device.
"""
import os

import torch

_DTYPES = {"fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16,
           "bfloat16": torch.bfloat16}
_compute_dtype = _DTYPES[os.environ.get("SEGMENTRON_HIP_DTYPE", "fp32").lower()]


def compute_dtype():
    """Element type of activations / packed weights inside the HIP path.  fp32 = parity path
    (exact-fp32 MFMA), bf16 = throughput path (fp32 accumulation, fp32 BN statistics, fp32
    master weights and logits).  SURVEY.md F9 / Appendix F."""
    return _compute_dtype


def set_compute_dtype(dtype):
    global _compute_dtype
    if isinstance(dtype, str):
        dtype = _DTYPES[dtype.lower()]
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("compute dtype must be float32 or bfloat16")
    _compute_dtype = dtype


from .config import cfg  # noqa: E402,F401
from .models import MODEL_REGISTRY, get_segmentation_model  # noqa: E402,F401
from . import torch_ops  # noqa: E402,F401  (registers torch.ops.segmentron_hip.*)
