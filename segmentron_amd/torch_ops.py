"""`torch.ops.segmentron_hip.*` — the C-ABI kernels as PyTorch custom operators
(north_star: "hand-written HIP kernels exposed as torch custom ops via a thin C-ABI extension";
SURVEY.md §8b last row).

Each operator is registered with `torch.library.custom_op` (schema inferred from the type hints),
a fake / meta implementation (`register_fake`: shapes and dtypes without touching the device, so
FakeTensorMode / torch.compile / torch.export can trace through), and `register_autograd` whose
backward is itself a custom operator.  Tensors are NHWC `[N, H, W, C]` HIP tensors (float32 or
bfloat16; channel-slice views of wider buffers are fine), parameters are float32 in torch's own
layouts — the same conventions as the C-ABI (include/segmentron_hip.h).

    y  = torch.ops.segmentron_hip.conv2d(x, weight, bias, stride, padding, dilation, relu_in)
    y  = torch.ops.segmentron_hip.depthwise_conv3x3(x, weight, stride, dilation, relu_in)
    y  = torch.ops.segmentron_hip.interpolate_bilinear(x, out_h, out_w, align_corners)
    lo = torch.ops.segmentron_hip.upsample_cross_entropy(logits, target, out_h, out_w,
                                                         ignore_index, align_corners)   # [2]
    out, att, raw = torch.ops.segmentron_hip.criss_cross_attention(q, k, v, x, gamma)
    cnt = torch.ops.segmentron_hip.segmentation_counts(logits_nchw, target, nclass)

The module tree (segmentron_amd.modules / .models) drives the same C-ABI wrappers
(segmentron_amd.hip_ops) directly through its deferred-BatchNorm autograd layer
(segmentron_amd.functional) — one Python frame per launch matters there; only the fused loss,
issued once per step, goes through `torch.ops` in the product path.  Importing this module is what
registers the operators (segmentron_amd/__init__.py does).
"""
from typing import Optional, Tuple

import torch

from . import hip_ops as K

_NS = "segmentron_hip"
_PRO_RELU = (K.PRO_RELU, None, None)


def _round_up(v, m):
    return (v + m - 1) // m * m


def _out(hi, k, stride, pad, dil):
    return (hi + 2 * pad - dil * (k - 1) - 1) // stride + 1


# ----------------------------------------------------------------------------- conv2d (groups=1)
@torch.library.custom_op(_NS + "::conv2d", mutates_args=())
def conv2d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], stride: int,
           padding: int, dilation: int, relu_in: bool) -> torch.Tensor:
    """nn.Conv2d(groups=1) on NHWC x with weight [O, C, KH, KW] fp32 (optionally relu(x) first):
    implicit GEMM on MFMA (seg_conv_gemm_fwd)."""
    from . import functional as F
    O, Cw, KH, KW = weight.shape
    wp = F.pack_conv_weight(weight, x.shape[-1], x.dtype)
    y, _ = K.conv_gemm(x, wp, O, KH, KW, stride, padding, dilation,
                       _PRO_RELU if relu_in else None, bias)
    return y


@conv2d.register_fake
def _(x, weight, bias, stride, padding, dilation, relu_in):
    N, H, W, _ = x.shape
    O, _, KH, KW = weight.shape
    return x.new_empty((N, _out(H, KH, stride, padding, dilation),
                        _out(W, KW, stride, padding, dilation), O))


@torch.library.custom_op(_NS + "::conv2d_backward", mutates_args=())
def conv2d_backward(x: torch.Tensor, dy: torch.Tensor, weight: torch.Tensor, stride: int,
                    padding: int, dilation: int, relu_in: bool, need_dx: bool
                    ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (dx NHWC like x (empty [0] if not need_dx), dweight [O,C,KH,KW] fp32, dbias [O] fp32)."""
    from . import functional as F
    O, Cw, KH, KW = weight.shape
    vec = K.vec_of(x.dtype)
    N, Ho, Wo, _ = dy.shape
    if O % vec != 0 or K.nhwc(dy)[4] % vec != 0:  # ragged output channels (e.g. 19 classes)
        dyp = torch.zeros((N, Ho, Wo, _round_up(O, vec)), dtype=dy.dtype, device=dy.device)
        dyp[..., :O] = dy
        dy_full, dy = dyp, dyp[..., :O]
    else:
        dy_full = dy
    Cx = x.shape[-1]
    pro = _PRO_RELU if relu_in else None
    dWp = K.conv_wgrad(x, dy, O, KH, KW, stride, padding, dilation, pro)
    dW = dWp.view(O, KH, KW, Cx)[..., :Cw].permute(0, 3, 1, 2).contiguous()
    db = K.bn_bwd_reduce(dy_full, dy_full, (K.PRO_NONE, None, None))[:O].float()
    if not need_dx:
        return x.new_empty((0,)), dW, db
    Op, dt = dy_full.shape[-1], x.dtype
    if stride == 1:
        wt = F.pack_conv_weight_dgrad(weight, Op, dt)
        g, _ = K.conv_gemm(dy_full, wt, Cw, KH, KW, 1, dilation * (KH - 1) - padding, dilation)
    elif KH == 1 and KW == 1 and padding == 0:
        wt = F.pack_conv_weight_dgrad(weight, Op, dt)
        g, _ = K.conv_gemm(dy_full, wt, Cw, 1, 1, 1, 0, 1, scatter=(x.shape[1], x.shape[2], stride))
    else:
        wt = F.pack_conv_weight_tconv(weight, Op, dt)
        g, _ = K.conv_gemm(dy_full, wt, Cw, KH, KW, stride, padding, dilation,
                           tconv_out_hw=(x.shape[1], x.shape[2]))
    if Cw != Cx:  # channel-padded input (the 3-channel image): gradient of the padding is zero
        gp = torch.zeros(x.shape, dtype=dt, device=x.device)
        gp[..., :Cw] = g
        g = gp
    if relu_in:
        g = K.bn_bwd_apply(g, x, _PRO_RELU, out=g)
    return g, dW, db


@conv2d_backward.register_fake
def _(x, dy, weight, stride, padding, dilation, relu_in, need_dx):
    dx = x.new_empty(x.shape) if need_dx else x.new_empty((0,))
    return (dx, weight.new_empty(weight.shape, dtype=torch.float32),
            weight.new_empty((weight.shape[0],), dtype=torch.float32))


def _conv2d_setup(ctx, inputs, output):
    x, weight, bias, stride, padding, dilation, relu_in = inputs
    ctx.save_for_backward(x, weight)
    ctx.cfg = (stride, padding, dilation, relu_in, bias is not None)


def _conv2d_bwd(ctx, dy):
    x, weight = ctx.saved_tensors
    stride, padding, dilation, relu_in, has_bias = ctx.cfg
    dx, dW, db = conv2d_backward(x, dy, weight, stride, padding, dilation, relu_in,
                                 ctx.needs_input_grad[0])
    return (dx if ctx.needs_input_grad[0] else None, dW, db if has_bias else None, None, None,
            None, None)


conv2d.register_autograd(_conv2d_bwd, setup_context=_conv2d_setup)


# ----------------------------------------------------------------------------- depthwise 3x3
@torch.library.custom_op(_NS + "::depthwise_conv3x3", mutates_args=())
def depthwise_conv3x3(x: torch.Tensor, weight: torch.Tensor, stride: int, dilation: int,
                      relu_in: bool) -> torch.Tensor:
    """nn.Conv2d(C, C, 3, stride, padding=dilation, dilation, groups=C) on NHWC x, weight
    [C,1,3,3] fp32 (segmentron/modules/basic.py:38-40)."""
    from . import functional as F
    w = weight if K.dw_tiled(stride, dilation) else F.pack_dw_weight(weight)
    y, _ = K.dwconv(x, w, stride, dilation, _PRO_RELU if relu_in else None)
    return y


@depthwise_conv3x3.register_fake
def _(x, weight, stride, dilation, relu_in):
    N, H, W, C = x.shape
    return x.new_empty((N, _out(H, 3, stride, dilation, dilation),
                        _out(W, 3, stride, dilation, dilation), C))


@torch.library.custom_op(_NS + "::depthwise_conv3x3_backward", mutates_args=())
def depthwise_conv3x3_backward(x: torch.Tensor, dy: torch.Tensor, weight: torch.Tensor,
                               stride: int, dilation: int, relu_in: bool
                               ) -> Tuple[torch.Tensor, torch.Tensor]:
    from . import functional as F
    C = weight.shape[0]
    pro = _PRO_RELU if relu_in else None
    if K.nhwc(dy)[4] % K.vec_of(dy.dtype) != 0:
        dy = dy.contiguous()
    tiled = K.dw_tiled(stride, dilation)
    if stride == 1 and tiled:  # one pass: masked data gradient + weight gradient
        g, dW, _ = K.dwconv_bwd_fused(x, dy, weight, dilation, pro, want_bn=False,
                                      torch_layout=True)
        return g, dW
    if tiled:
        dW = K.dwconv_wgrad(x, dy, stride, dilation, pro, torch_layout=True)
        w = weight
    else:
        dW = K.dwconv_wgrad(x, dy, stride, dilation, pro).t().reshape(C, 1, 3, 3).contiguous()
        w = F.pack_dw_weight(weight, flipped=stride == 1)
    g = K.dwconv_dgrad(dy, w, stride, dilation, (x.shape[1], x.shape[2]))
    if relu_in:
        g = K.bn_bwd_apply(g, x, _PRO_RELU, out=g)
    return g, dW


@depthwise_conv3x3_backward.register_fake
def _(x, dy, weight, stride, dilation, relu_in):
    return x.new_empty(x.shape), weight.new_empty(weight.shape, dtype=torch.float32)


def _dw_setup(ctx, inputs, output):
    x, weight, stride, dilation, relu_in = inputs
    ctx.save_for_backward(x, weight)
    ctx.cfg = (stride, dilation, relu_in)


def _dw_bwd(ctx, dy):
    x, weight = ctx.saved_tensors
    dx, dW = depthwise_conv3x3_backward(x, dy, weight, *ctx.cfg)
    return dx, dW, None, None, None


depthwise_conv3x3.register_autograd(_dw_bwd, setup_context=_dw_setup)


# ----------------------------------------------------------------------------- bilinear resize
@torch.library.custom_op(_NS + "::interpolate_bilinear", mutates_args=())
def interpolate_bilinear(x: torch.Tensor, out_h: int, out_w: int, align_corners: bool
                         ) -> torch.Tensor:
    """F.interpolate(mode='bilinear', align_corners) on NHWC x (deeplabv3_plus.py:39,71)."""
    return K.bilinear(x, (out_h, out_w), None, None, align_corners)


@interpolate_bilinear.register_fake
def _(x, out_h, out_w, align_corners):
    return x.new_empty((x.shape[0], out_h, out_w, x.shape[3]))


@torch.library.custom_op(_NS + "::interpolate_bilinear_backward", mutates_args=())
def interpolate_bilinear_backward(dy: torch.Tensor, in_h: int, in_w: int, align_corners: bool
                                  ) -> torch.Tensor:
    if K.nhwc(dy)[4] % K.vec_of(dy.dtype) != 0:
        dy = dy.contiguous()
    return K.bilinear_bwd(dy, (in_h, in_w), align_corners)


@interpolate_bilinear_backward.register_fake
def _(dy, in_h, in_w, align_corners):
    return dy.new_empty((dy.shape[0], in_h, in_w, dy.shape[3]))


def _bil_setup(ctx, inputs, output):
    x, out_h, out_w, align = inputs
    ctx.cfg = (x.shape[1], x.shape[2], align)


def _bil_bwd(ctx, dy):
    return interpolate_bilinear_backward(dy, *ctx.cfg), None, None, None


interpolate_bilinear.register_autograd(_bil_bwd, setup_context=_bil_setup)


# ----------------------------------------------------------------------------- fused loss tail
@torch.library.custom_op(_NS + "::upsample_cross_entropy", mutates_args=())
def upsample_cross_entropy(logits: torch.Tensor, target: torch.Tensor, out_h: int, out_w: int,
                           ignore_index: int, align_corners: bool) -> torch.Tensor:
    """F.cross_entropy(F.interpolate(logits, (out_h, out_w), 'bilinear', align_corners), target,
    ignore_index) fused on the low-resolution NHWC logits.  -> float32[2]: (mean loss over the
    valid pixels, 1 / number of valid pixels)."""
    return K.upsample_ce_fwd(logits, target, (out_h, out_w), ignore_index, align_corners)


@upsample_cross_entropy.register_fake
def _(logits, target, out_h, out_w, ignore_index, align_corners):
    return logits.new_empty((2,), dtype=torch.float32)


@torch.library.custom_op(_NS + "::upsample_cross_entropy_backward", mutates_args=())
def upsample_cross_entropy_backward(logits: torch.Tensor, target: torch.Tensor,
                                    loss_out: torch.Tensor, grad: torch.Tensor, out_h: int,
                                    out_w: int, ignore_index: int, align_corners: bool
                                    ) -> torch.Tensor:
    C = logits.shape[-1]
    pitch = _round_up(C, K.vec_of(logits.dtype))
    d = K.upsample_ce_bwd(logits, target, (out_h, out_w), ignore_index, loss_out, grad, pitch,
                          align_corners)
    return d if pitch == C else d[..., :C].contiguous()


@upsample_cross_entropy_backward.register_fake
def _(logits, target, loss_out, grad, out_h, out_w, ignore_index, align_corners):
    return logits.new_empty(logits.shape)


def _uce_setup(ctx, inputs, output):
    logits, target, out_h, out_w, ignore_index, align = inputs
    ctx.save_for_backward(logits, target, output)
    ctx.cfg = (out_h, out_w, ignore_index, align)


def _uce_bwd(ctx, g):
    logits, target, out = ctx.saved_tensors
    # only element 0 (the loss) carries gradient
    return (upsample_cross_entropy_backward(logits, target, out, g[0].reshape(1).contiguous(),
                                            *ctx.cfg), None, None, None, None, None)


upsample_cross_entropy.register_autograd(_uce_bwd, setup_context=_uce_setup)

# ----------------------------------------------------------------------------- criss-cross attention
@torch.library.custom_op(_NS + "::criss_cross_attention", mutates_args=())
def criss_cross_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, x: torch.Tensor,
                          gamma: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """gamma * CCA(q, k, v) + x on NHWC tensors (cc_attention.py:60-72).  -> (out, attention
    fp32 [N,H,W,W+H-1], un-scaled aggregation) — the last two are what backward needs."""
    g32 = gamma.detach().float().contiguous()
    att = K.cca_attention(q, k)
    out, raw = K.cca_map(att, v, gamma=g32, res=x, want_raw=True)
    return out, att, raw


@criss_cross_attention.register_fake
def _(q, k, v, x, gamma):
    N, H, W, _ = v.shape
    return (v.new_empty(v.shape), v.new_empty((N, H, W, H + W - 1), dtype=torch.float32),
            v.new_empty(v.shape))


@torch.library.custom_op(_NS + "::criss_cross_attention_backward", mutates_args=())
def criss_cross_attention_backward(dout: torch.Tensor, q: torch.Tensor, k: torch.Tensor,
                                   v: torch.Tensor, att: torch.Tensor, raw: torch.Tensor,
                                   gamma: torch.Tensor
                                   ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (dq, dk, dv, dgamma[1]); dx is dout itself."""
    g32 = gamma.detach().float().contiguous()
    dout = dout.contiguous()
    C = dout.shape[-1]
    dgamma = K.bn_bwd_reduce(dout, raw, (K.PRO_NONE, None, None))[C:2 * C].sum().float().view(1)
    de = K.cca_attention_bwd(dout, v, att, g32)
    return (K.cca_map(de, k), K.cca_map(de, q, transposed=True),
            K.cca_map(att, dout, transposed=True, gamma=g32), dgamma)


@criss_cross_attention_backward.register_fake
def _(dout, q, k, v, att, raw, gamma):
    return (q.new_empty(q.shape), k.new_empty(k.shape), v.new_empty(v.shape),
            gamma.new_empty((1,), dtype=torch.float32))


def _cca_setup(ctx, inputs, output):
    q, k, v, x, gamma = inputs
    ctx.save_for_backward(q, k, v, output[1], output[2], gamma)


def _cca_bwd(ctx, dout, datt, draw):
    q, k, v, att, raw, gamma = ctx.saved_tensors
    dq, dk, dv, dg = criss_cross_attention_backward(dout, q, k, v, att, raw, gamma)
    return dq, dk, dv, dout, dg.to(gamma.dtype)


criss_cross_attention.register_autograd(_cca_bwd, setup_context=_cca_setup)


# ----------------------------------------------------------------------------- metric counters
@torch.library.custom_op(_NS + "::segmentation_counts", mutates_args=())
def segmentation_counts(logits: torch.Tensor, target: torch.Tensor, nclass: int) -> torch.Tensor:
    """pixAcc / mIoU counts of fp32 NCHW logits (utils/score.py:83-113) -> int64 [2 + 3*nclass]:
    correct, labelled, inter[], pred[], lab[]."""
    return K.metric_update_nchw(logits, target, nclass, K.metric_counters(nclass, logits.device))


@segmentation_counts.register_fake
def _(logits, target, nclass):
    return logits.new_empty((2 + 3 * nclass,), dtype=torch.int64)


OPS = ("conv2d", "conv2d_backward", "depthwise_conv3x3", "depthwise_conv3x3_backward",
       "interpolate_bilinear", "interpolate_bilinear_backward", "upsample_cross_entropy",
       "upsample_cross_entropy_backward", "criss_cross_attention",
       "criss_cross_attention_backward", "segmentation_counts")
