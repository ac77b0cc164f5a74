"""Stand-in for `thop.profile` as used by segmentron/utils/visualize.py:36-40
(`flops, params = profile(model, inputs=(input,), verbose=False)`).

thop counts multiply-accumulates with forward hooks on nn.Conv2d & co.; the MI355X models never
call those modules' forward (their kernels read the parameters directly), so here the MACs are
counted where they are issued: every convolution launch of segmentron_amd.hip_ops during ONE
forward pass (2-D convolutions only, like the survey's census).  Plain torch
models fall back to hooks on nn.Conv2d / nn.Linear."""
import torch
import torch.nn as nn


def _hook_macs(model, inputs):
    macs, hooks = [0], []

    def conv_hook(m, i, o):
        k = m.kernel_size[0] * m.kernel_size[1] * (m.in_channels // m.groups)
        macs[0] += o.numel() * k

    def lin_hook(m, i, o):
        macs[0] += o.numel() * m.in_features

    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            hooks.append(m.register_forward_hook(conv_hook))
        elif isinstance(m, nn.Linear):
            hooks.append(m.register_forward_hook(lin_hook))
    try:
        with torch.no_grad():
            model(*inputs)
    finally:
        for h in hooks:
            h.remove()
    return macs[0]


def _hip_macs(model, inputs):
    from segmentron_amd import hip_ops as K
    macs = [0]
    orig_gemm, orig_dw = K.conv_gemm, K.dwconv

    def gemm(x, w_packed, O, KH, KW, stride, pad, dil, *a, **kw):
        y, p = orig_gemm(x, w_packed, O, KH, KW, stride, pad, dil, *a, **kw)
        macs[0] += y.shape[0] * y.shape[1] * y.shape[2] * O * (w_packed.shape[1])
        return y, p

    def dw(x, w9c, stride, dil, *a, **kw):
        y, p = orig_dw(x, w9c, stride, dil, *a, **kw)
        macs[0] += y.numel() * 9
        return y, p

    K.conv_gemm, K.dwconv = gemm, dw
    try:
        with torch.no_grad():
            model(*inputs)
    finally:
        K.conv_gemm, K.dwconv = orig_gemm, orig_dw
    return macs[0]


def profile(model, inputs=(), verbose=False, **kwargs):
    """-> (MACs of one forward pass, number of parameters)."""
    params = sum(p.numel() for p in model.parameters())
    is_hip = type(model).__module__.startswith(("segmentron_amd", "segmentron."))
    macs = _hip_macs(model, inputs) if is_hip else _hook_macs(model, inputs)
    return float(macs), float(params)
