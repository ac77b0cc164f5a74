#!/usr/bin/env python3
"""Fixtures for SegBaseModel.evaluate (segmentron/models/segbase.py:44-79: multi-scale / flip /
pad / crop inference glue), generated from the REFERENCE in the dev container.

The reference method is called UNBOUND on a stub whose `forward` is a fixed, position-dependent
torch function, so the fixture pins the image-space glue only (resize, the F.pad argument order
of _pad_image, crop, flip, score accumulation) — the network itself is pinned elsewhere.

    python oracle/gen_golden_eval.py      -> tests/golden/evaluate_cases.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_import  # noqa: E402

CASES = [  # name, image (h, w), TEST.SCALES, TEST.FLIP, TEST.CROP_SIZE
    ("single", (37, 65), [1.0], False, None),
    ("multiscale_flip", (37, 65), [0.5, 1.0, 1.5], True, None),
    ("crop_tuple", (37, 65), [1.0, 1.25], True, (40, 72)),
    ("crop_int_portrait", (61, 45), [0.75, 1.0], False, 64),
]


def stub_forward(x):
    """[N,3,H,W] -> ([N,5,H,W],): channel mix + absolute-position ramps (catches a wrong crop,
    flip or pad side)."""
    n, _, h, w = x.shape
    mix = torch.tensor([[1.0, -0.5, 0.25], [0.3, 0.3, 0.3], [-1.0, 0.7, 0.1], [0.0, 1.0, -1.0],
                        [0.5, 0.5, -0.25]], dtype=x.dtype)
    y = torch.einsum("oc,nchw->nohw", mix, x)
    ramp_w = torch.arange(w, dtype=x.dtype).view(1, 1, 1, w) * 0.01
    ramp_h = torch.arange(h, dtype=x.dtype).view(1, 1, h, 1) * 0.02
    scale = torch.arange(1, 6, dtype=x.dtype).view(1, 5, 1, 1)
    return (y + scale * (ramp_w - ramp_h),)


class Stub:
    aux = False

    def forward(self, x):
        return stub_forward(x)


def image(h, w):
    g = torch.Generator().manual_seed(h * 1000 + w)
    return torch.randn(2, 3, h, w, generator=g)


def main():
    ref_import._install_stubs()
    sys.path.insert(0, ref_import.REFERENCE_ROOT)
    for name in [n for n in sys.modules if n == "segmentron" or n.startswith("segmentron.")]:
        del sys.modules[name]
    from segmentron.config import cfg
    from segmentron.models.segbase import SegBaseModel
    out = {}
    for name, (h, w), scales, flip, crop in CASES:
        cfg.TEST.SCALES, cfg.TEST.FLIP, cfg.TEST.CROP_SIZE = scales, flip, crop
        with torch.no_grad():
            out[name] = SegBaseModel.evaluate(Stub(), image(h, w)).numpy()
        print(name, out[name].shape, float(np.abs(out[name]).max()))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "evaluate_cases.npz"), **out)


if __name__ == "__main__":
    main()
