"""Generates tests/golden/cca_ref_vectors.npz with the REFERENCE's criss-cross-attention kernels
(ca_cuda.cu:8-177) compiled as host C++ (oracle/cca_ref/build.sh -> oracle/_ref/libcca_ref.so):
for seeded inputs the four `_C` entry points' outputs — energies, aggregation and the four
backward kernels' gradients.  tests/test_cca.py pins oracle/torch_ref.py's cca_weight / cca_map
(and torch autograd of them) to these vectors, so the oracle is anchored to numbers the
reference's own code produced, not to a transcription of it.

    python oracle/gen_golden_cca.py        (needs /root/reference; here, not on the GPU box)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cca_ref, torch_ref as R  # noqa: E402

SHAPES = [(2, 3, 4, 5), (1, 2, 5, 3), (1, 1, 1, 4), (1, 2, 3, 1), (1, 2, 33, 35), (1, 3, 40, 7)]


def inputs(shape, dtype):
    n, c, h, w = shape
    g = torch.Generator().manual_seed(1000 + n * 7 + c * 5 + h * 3 + w)
    mk = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64).to(dtype)
    t, f = mk(n, c, h, w), mk(n, c, h, w)
    v = mk(n, 2 * c, h, w)
    att = torch.softmax(mk(n, h + w - 1, h, w), 1)
    dwt, dout = mk(n, h + w - 1, h, w), mk(n, 2 * c, h, w)
    return t, f, v, att, dwt, dout


def main():
    assert cca_ref.available(), "build oracle/_ref/libcca_ref.so first (oracle/cca_ref/build.sh)"
    out = {"shapes": np.array(SHAPES)}
    for i, shape in enumerate(SHAPES):
        for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            t, f, v, att, dwt, dout = inputs(shape, dt)
            weight = cca_ref.ca_forward(t, f)
            dt_, df_ = cca_ref.ca_backward(dwt, t, f)
            agg = cca_ref.ca_map_forward(att, v)
            dw_, dg_ = cca_ref.ca_map_backward(dout, att, v)
            for k, val in (("weight", weight), ("dt", dt_), ("df", df_), ("agg", agg),
                           ("dw", dw_), ("dg", dg_)):
                out["%d_%s_%s" % (i, tag, k)] = val.numpy()
            if dt == torch.float64:  # the oracle restatement agrees with what it is pinned to
                assert (R.cca_weight(t, f) - weight).abs().max() < 1e-12
                assert (R.cca_map(att, v) - agg).abs().max() < 1e-12
    path = os.path.join(ROOT, "tests", "golden", "cca_ref_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "(%d arrays, %d bytes)" % (len(out), os.path.getsize(path)))


if __name__ == "__main__":
    main()
