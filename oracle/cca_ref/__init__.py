"""TEST INFRASTRUCTURE ONLY: ctypes binding of oracle/_ref/libcca_ref.so — the reference's own
criss-cross-attention kernels (segmentron/modules/csrc/criss_cross_attention/ca_cuda.cu:8-177)
compiled as host C++ by oracle/cca_ref/build.sh.  Same four entry points as the reference's
`_C` extension (ca.h:25-73): ca_forward, ca_backward, ca_map_forward, ca_map_backward, on
contiguous NCHW float32 / float64 CPU tensors."""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(_HERE, "..", "_ref", "libcca_ref.so")
_lib = None


def available(build=True):
    """True if the compiled reference can be used here (builds it when the reference checkout is
    present and the .so is not)."""
    global _lib
    if _lib is not None:
        return True
    if not os.path.exists(SO) and build:
        try:
            subprocess.check_call(["sh", os.path.join(_HERE, "build.sh")],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        except (subprocess.CalledProcessError, OSError):
            return False
    if not os.path.exists(SO):
        return False
    _lib = ctypes.CDLL(SO)
    return True


def _fn(name, t):
    assert available(), "oracle/_ref/libcca_ref.so is not built (oracle/cca_ref/build.sh)"
    sfx = {torch.float32: "f32", torch.float64: "f64"}[t.dtype]
    f = getattr(_lib, "cca_ref_%s_%s" % (name, sfx))
    f.restype = None
    return f


def _p(t):
    assert t.is_contiguous() and t.device.type == "cpu"
    return ctypes.c_void_p(t.data_ptr())


def ca_forward(t, f):
    """ca_forward_cuda (ca_cuda.cu:184-215): energies [N, H+W-1, H, W]."""
    t, f = t.contiguous(), f.contiguous()
    n, c, h, w = t.shape
    weight = torch.empty((n, h + w - 1, h, w), dtype=t.dtype)
    _fn("forward", t)(_p(t), _p(f), _p(weight), n, c, h, w)
    return weight


def ca_backward(dw, t, f):
    """ca_backward_cuda (ca_cuda.cu:217-260): (dt, df)."""
    dw, t, f = dw.contiguous(), t.contiguous(), f.contiguous()
    n, c, h, w = t.shape
    dt, df = torch.empty_like(t), torch.empty_like(f)
    _fn("backward", t)(_p(dw), _p(t), _p(f), _p(dt), _p(df), n, c, h, w)
    return dt, df


def ca_map_forward(weight, g):
    """ca_map_forward_cuda (ca_cuda.cu:262-290): aggregation [N, C, H, W]."""
    weight, g = weight.contiguous(), g.contiguous()
    n, c, h, w = g.shape
    out = torch.empty_like(g)
    _fn("map_forward", g)(_p(weight), _p(g), _p(out), n, c, h, w)
    return out


def ca_map_backward(dout, weight, g):
    """ca_map_backward_cuda (ca_cuda.cu:292-335): (dw, dg)."""
    dout, weight, g = dout.contiguous(), weight.contiguous(), g.contiguous()
    n, c, h, w = g.shape
    dw, dg = torch.empty_like(weight), torch.empty_like(g)
    _fn("map_backward", g)(_p(dout), _p(weight), _p(g), _p(dw), _p(dg), n, c, h, w)
    return dw, dg
