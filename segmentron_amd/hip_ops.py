"""Thin tensor-level wrappers over the C-ABI (include/segmentron_hip.h).

Everything here takes raw ``torch`` CUDA(HIP) tensors purely as device-memory handles — NHWC views
``[N, H, W, C]`` whose channel dimension may be a slice of a wider buffer — and launches on the
current HIP stream.  No autograd, no fallbacks: a CPU tensor raises.
"""
import torch

from ._lib import LIB

PRO_NONE, PRO_RELU, PRO_AFFINE, PRO_AFFINE_RELU = 0, 1, 2, 3
_DT = {torch.float32: 0, torch.bfloat16: 1}


def vec_of(dtype):
    return 8 if dtype == torch.bfloat16 else 4


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream.  The raw getter is ~10x cheaper than building a
    torch.cuda.Stream object — with ~1200 launches per step that was 4 ms of host time in the
    launch-bound eager path (N > 1 DDP runs)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


def _count(count):
    """BatchNorm element count: a host number, or (SyncBN) a device float64 scalar tensor that
    came out of the statistics all-reduce -> (host double, device pointer)."""
    if isinstance(count, torch.Tensor):
        return 0.0, count.data_ptr()
    return float(count), 0


def nhwc(t):
    """-> (N, H, W, C, ld) of an NHWC view; checks it is addressable as rows with pitch ld."""
    if not t.is_cuda:
        raise RuntimeError("segmentron_amd ops need HIP device tensors (no CPU fallback)")
    if t.dim() != 4:
        raise RuntimeError("expected NHWC 4-d tensor, got %s" % (tuple(t.shape),))
    N, H, W, C = t.shape
    # torch leaves arbitrary strides on size-1 dims: derive the row pitch from the innermost
    # dimension that actually has extent
    if W > 1:
        ld = t.stride(2)
    elif H > 1:
        ld = t.stride(1)
    elif N > 1:
        ld = t.stride(0)
    else:
        ld = C
    ok = (t.stride(3) == 1 or C == 1) and ld >= C
    if W > 1 and H > 1:
        ok = ok and t.stride(1) == W * ld
    if N > 1 and (H > 1 or W > 1):
        ok = ok and t.stride(0) == H * W * ld
    if not ok:
        raise RuntimeError("tensor is not a dense-row NHWC view: shape %s strides %s"
                           % (tuple(t.shape), t.stride()))
    return N, H, W, C, ld


def _pro(pro):
    if pro is None:
        return PRO_NONE, None, None
    return pro


def new_act(N, H, W, C, dtype, device, pitch=None):
    """Allocate an NHWC activation [N,H,W,C] (optionally as the leading slice of pitch channels)."""
    if pitch is None or pitch == C:
        return torch.empty((N, H, W, C), dtype=dtype, device=device)
    return torch.empty((N, H, W, pitch), dtype=dtype, device=device)[..., :C]


# ----------------------------------------------------------------------------- conv (GEMM)
def conv_out_size(Hi, k, stride, pad, dil):
    return (Hi + 2 * pad - dil * (k - 1) - 1) // stride + 1


def conv_gemm(x, w_packed, O, KH, KW, stride, pad, dil, pro=None, bias=None, out=None,
              want_stats=False, scatter=None, ep=None, tconv_out_hw=None):
    """x NHWC; w_packed [O, KH*KW*C] in x.dtype.  Returns (y, stat_partial|None).
    scatter=(out_H, out_W, s): write output pixel (ho,wo) at (ho*s, wo*s) of a zero-filled
    [N,out_H,out_W,O] tensor (data gradient of a strided 1x1 conv)."""
    N, Hi, Wi, C, ldx = nhwc(x)
    if tconv_out_hw is not None:  # transposed-stride gather: x is dy, the output is dx
        Ho, Wo = tconv_out_hw
    else:
        Ho, Wo = conv_out_size(Hi, KH, stride, pad, dil), conv_out_size(Wi, KW, stride, pad, dil)
    mode, ps, pt = _pro(pro)
    if scatter is None:
        oH, oW, os_ = Ho, Wo, 1
        if out is None:
            out = torch.empty((N, Ho, Wo, O), dtype=x.dtype, device=x.device)
    else:
        oH, oW, os_ = scatter
        if out is None:
            out = torch.zeros((N, oH, oW, O), dtype=x.dtype, device=x.device)
    ldy = nhwc(out)[4]
    partial = None
    if want_stats:
        tm = LIB.query("seg_conv_gemm_stat_rows", _DT[x.dtype], N, Ho, Wo, C, O, KH, KW, stride,
                       pad, dil, 1 if tconv_out_hw is not None else 0, int(bias is not None), mode)
        partial = torch.empty((tm, 2, O), dtype=torch.float32, device=x.device)
    assert w_packed.dtype == x.dtype and w_packed.is_contiguous()
    ep_x, ldep, ep_c0, ep_c1 = None, 0, None, None
    if ep is not None:  # (x_like_output, c0, c1): y = acc - c0 - c1 * x  (folded-BN backward)
        ep_x, ep_c0, ep_c1 = ep
        assert tuple(ep_x.shape) == tuple(out.shape)
        ldep = nhwc(ep_x)[4]
    LIB.call("seg_conv_gemm_fwd", _DT[x.dtype], _p(x), ldx, N, Hi, Wi, C, _p(w_packed), O, KH, KW,
             stride, pad, dil, mode, _p(ps), _p(pt), _p(bias), _p(out), ldy, Ho, Wo, oH, oW, os_,
             _p(partial), _p(ep_x), ldep, _p(ep_c0), _p(ep_c1),
             1 if tconv_out_hw is not None else 0, _stream())
    return out, partial


def colsum(partial2d, f64=True):
    """partial2d: fp32 [R, L] -> [L] (float64 or float32)."""
    R, L = partial2d.shape
    dev = partial2d.device
    out = torch.empty(L, dtype=torch.float64 if f64 else torch.float32, device=dev)
    ws = torch.empty(64 * L, dtype=torch.float64, device=dev) if R > 128 else None
    LIB.call("seg_colsum", _p(partial2d), R, L, _p(out) if f64 else 0, 0 if f64 else _p(out),
             _p(ws), _stream())
    return out


# ---- nn.GroupNorm (csrc/groupnorm.hip) ----------------------------------------------------------
def gn_moments(u, v=None):
    """NHWC u (, v) -> fp32 [N, chunks, 2, C]: (sum u, sum u*v) per sample, pixel chunk and channel
    (v None: sum u, sum u^2)."""
    N, H, W, C, ldu = nhwc(u)
    ldv = 0
    if v is not None:
        assert tuple(v.shape) == tuple(u.shape) and v.dtype == u.dtype
        ldv = nhwc(v)[4]
    chunks = LIB.query("seg_gn_chunks", H * W)
    partial = torch.empty((N, chunks, 2, C), dtype=torch.float32, device=u.device)
    LIB.call("seg_gn_moments", _DT[u.dtype], _p(u), ldu, _p(v), ldv, N, H * W, C, _p(partial),
             _stream())
    return partial


def gn_fwd_finalize(partial, HW, G, gamma, beta, eps):
    """-> (mean_rstd fp32 [N, G, 2], coef fp32 [N, 3, C]) with z = coef[n,0,c]*x + coef[n,2,c]."""
    N, _, _, C = partial.shape
    mean_rstd = torch.empty((N, G, 2), dtype=torch.float32, device=partial.device)
    coef = torch.empty((N, 3, C), dtype=torch.float32, device=partial.device)
    LIB.call("seg_gn_fwd_finalize", _p(partial), N, HW, C, G, _p(gamma), _p(beta), float(eps),
             _p(mean_rstd), _p(coef), _stream())
    return mean_rstd, coef


def gn_bwd_finalize(partial, HW, G, mean_rstd, gamma):
    """moments of (dz, x) -> (coef [N, 3, C] of dx = k1*dz + k2*x + k3, contrib [N, 2, C]: the
    per-sample terms of dgamma | dbeta)."""
    N, _, _, C = partial.shape
    coef = torch.empty((N, 3, C), dtype=torch.float32, device=partial.device)
    contrib = torch.empty((N, 2, C), dtype=torch.float32, device=partial.device)
    LIB.call("seg_gn_bwd_finalize", _p(partial), N, HW, C, G, _p(mean_rstd), _p(gamma), _p(coef),
             _p(contrib), _stream())
    return coef, contrib


def gn_affine(u, v, coef, out=None):
    """out = coef[n,0,c]*u + coef[n,1,c]*v + coef[n,2,c]  (v may be None)."""
    N, H, W, C, ldu = nhwc(u)
    ldv = 0 if v is None else nhwc(v)[4]
    if out is None:
        out = torch.empty((N, H, W, C), dtype=u.dtype, device=u.device)
    LIB.call("seg_gn_affine", _DT[u.dtype], _p(u), ldu, _p(v), ldv, _p(coef), _p(out),
             nhwc(out)[4], N, H * W, C, _stream())
    return out


def colsum_count(partial2d, count):
    """partial2d fp32 [R, L] -> float64 [L + 1] = (column sums | count): the SyncBatchNorm
    forward message of one BatchNorm in one launch."""
    R, L = partial2d.shape
    dev = partial2d.device
    out = torch.empty(L + 1, dtype=torch.float64, device=dev)
    ws = torch.empty(64 * L, dtype=torch.float64, device=dev) if R > 128 else None
    LIB.call("seg_colsum_count", _p(partial2d), R, L, _p(out), float(count), _p(ws), _stream())
    return out


def conv_wgrad(x, dy, O, KH, KW, stride, pad, dil, pro=None, raw_partial=False):
    """-> dW fp32 [O, KH*KW*C]."""
    N, Hi, Wi, C, ldx = nhwc(x)
    Nd, Ho, Wo, Od, lddy = nhwc(dy)
    assert Nd == N and Od == O
    mode, ps, pt = _pro(pro)
    K = KH * KW * C
    splits = LIB.query("seg_conv_gemm_wgrad_splits", _DT[x.dtype], N, Ho, Wo, C, O, KH, KW,
                       stride, pad, dil, mode)
    partial = torch.empty((splits, O * K), dtype=torch.float32, device=x.device)
    LIB.call("seg_conv_gemm_wgrad", _DT[x.dtype], _p(x), ldx, N, Hi, Wi, C, _p(dy), lddy, Ho, Wo,
             O, KH, KW, stride, pad, dil, mode, _p(ps), _p(pt), _p(partial), splits, _stream())
    if raw_partial:  # [splits, O*K]: the consumer sums the splits itself (fold_bwd_reduce)
        return partial
    if splits == 1:
        return partial.view(O, K)
    return colsum(partial, f64=False).view(O, K)


# ----------------------------------------------------------------------------- depthwise
def dw_tiled(stride, dil):
    """True where the LDS-tiled depthwise kernels run: they read the taps in torch's own
    [C,1,3,3] layout (no repacking) and, for the data gradient, reversed in place."""
    return stride == 1 and dil in (1, 2)


def _dw_weight_arg(w, C, stride, dil, flip):
    """-> (tensor, w_layout).  [C,1,3,3]/[C,9] parameter tensors go straight to the tiled
    kernels; the strip kernels take the tap-major [9, C] packing."""
    if w.dim() == 4 or (w.dim() == 2 and w.shape == (C, 9) and C != 9):
        if not dw_tiled(stride, dil):
            raise ValueError("depthwise strip kernels need tap-major [9, C] weights")
        return w, 1 | (2 if flip else 0)
    return w, (2 if flip else 0) if dw_tiled(stride, dil) else 0


def dwconv(x, w9c, stride, dil, pro=None, out=None, want_stats=False):
    """w9c: tap-major [9, C] fp32, or (stride 1, dil <= 2) the [C,1,3,3] parameter itself."""
    N, Hi, Wi, C, ldx = nhwc(x)
    Ho, Wo = conv_out_size(Hi, 3, stride, dil, dil), conv_out_size(Wi, 3, stride, dil, dil)
    mode, ps, pt = _pro(pro)
    if out is None:
        out = torch.empty((N, Ho, Wo, C), dtype=x.dtype, device=x.device)
    ldy = nhwc(out)[4]
    w, layout = _dw_weight_arg(w9c, C, stride, dil, False)
    gy = LIB.query("seg_dwconv_grid_y", _DT[x.dtype], C, N, Ho, Wo, stride, dil, 0)
    partial = torch.empty((gy, 2, C), dtype=torch.float32, device=x.device) if want_stats else None
    LIB.call("seg_dwconv3x3", _DT[x.dtype], 0, _p(x), ldx, N, Hi, Wi, C, _p(w), layout, stride,
             dil, mode, _p(ps), _p(pt), _p(out), ldy, Ho, Wo, _p(partial), gy, _stream())
    return out, partial


def dwconv_dgrad(dy, w9c, stride, dil, in_hw, flipped=True):
    """stride 1: the forward correlation with the taps reversed.  `w9c` is either the
    [C,1,3,3] parameter (reversed inside the kernel), or a tap-major [9, C] packing that is
    ALREADY reversed (`flipped=True`, pack_dw_weight(..., flipped=True)) / not yet."""
    N, Ho, Wo, C, lddy = nhwc(dy)
    Hi, Wi = in_hw
    dx = torch.empty((N, Hi, Wi, C), dtype=dy.dtype, device=dy.device)
    gy = LIB.query("seg_dwconv_grid_y", _DT[dy.dtype], C, N, Hi, Wi, stride, dil, 0)
    if stride == 1:
        if w9c.dim() == 4:
            w, layout = w9c, 3
        elif dw_tiled(stride, dil):
            w, layout = w9c, 0 if flipped else 2
        else:
            w, layout = (w9c if flipped else w9c.flip(0).contiguous()), 0
        LIB.call("seg_dwconv3x3", _DT[dy.dtype], 0, _p(dy), lddy, N, Ho, Wo, C, _p(w), layout, 1,
                 dil, PRO_NONE, 0, 0, _p(dx), C, Hi, Wi, 0, gy, _stream())
    else:
        LIB.call("seg_dwconv3x3", _DT[dy.dtype], 1, _p(dy), lddy, N, Ho, Wo, C, _p(w9c), 0, stride,
                 dil, PRO_NONE, 0, 0, _p(dx), C, Hi, Wi, 0, gy, _stream())
    return dx


def dw_wgrad_finalize(pw, C):
    """weight-gradient partials [R, 9C] -> dW [C,1,3,3]."""
    dW = torch.empty((C, 1, 3, 3), dtype=torch.float32, device=pw.device)
    LIB.call("seg_dwconv3x3_wgrad_finalize", _p(pw), pw.shape[0], C, _p(dW), _stream())
    return dW


def dw_bwd_finalize(pb, pw, count, mean, invstd, gamma):
    """Both reductions behind a fused depthwise backward in one launch: BatchNorm-backward
    partials pb [Rb, 2C] -> (dgamma, dbeta, c0, c1) and weight-gradient partials pw [Rw, 9C] ->
    dW [C,1,3,3]."""
    C = mean.numel()
    out = torch.empty((4, C), dtype=torch.float32, device=mean.device)
    dW = torch.empty((C, 1, 3, 3), dtype=torch.float32, device=mean.device)
    LIB.call("seg_dw_bwd_finalize", _p(pb), pb.shape[0], float(count), _p(mean), _p(invstd),
             _p(gamma), _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]), _p(pw), pw.shape[0],
             _p(dW), C, _stream())
    return out[0], out[1], out[2], out[3], dW


def dw_bwd_finalize_sync(box, pb, pw, count_dev, mean, invstd, gamma, grad_scale):
    """dw_bwd_finalize for SyncBatchNorm: the BatchNorm sums are exchanged between the ranks
    inside the kernel (box: xgmi.PeerMailbox); the weight gradient stays local."""
    C = mean.numel()
    out = torch.empty((4, C), dtype=torch.float32, device=mean.device)
    dW = torch.empty((C, 1, 3, 3), dtype=torch.float32, device=mean.device)
    LIB.call("seg_dw_bwd_finalize_sync", box.handle, _p(pb), pb.shape[0], _p(count_dev), _p(mean),
             _p(invstd), _p(gamma), _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]), _p(pw),
             pw.shape[0], _p(dW), C, float(grad_scale), _stream())
    return out[0], out[1], out[2], out[3], dW


def dwconv_bwd_fused_add_ok(x, dil):
    """Can the fused depthwise backward add a second gradient in its store path?"""
    return bool(LIB.query("seg_dwconv3x3_bwd_fused_add_ok", int(dil)))


def dwconv_bwd_fused(x, dy, w, dil, pro=None, want_bn=False, torch_layout=False, raw_dw=False,
                     res=None):
    """stride-1 depthwise backward in one pass: returns (g masked by the prologue's ReLU,
    dW fp32 [9, C] (or [C,1,3,3] with torch_layout), bn_partial fp32 [gy, 2C] | None).
    w: tap-major [9, C] or (dil <= 2) the [C,1,3,3] parameter itself."""
    N, H, W, C, ldx = nhwc(x)
    lddy = nhwc(dy)[4]
    mode, ps, pt = _pro(pro)
    g = torch.empty((N, H, W, C), dtype=x.dtype, device=x.device)
    tiled = dw_tiled(1, dil)
    layout = 1 if w.dim() == 4 else 0
    if layout and not tiled:
        raise ValueError("depthwise strip kernels need tap-major [9, C] weights")
    gy = LIB.query("seg_dwconv_grid_y", _DT[x.dtype], C, N, H, W, 1, dil, 1)
    pw = torch.empty((gy, 9 * C), dtype=torch.float32, device=x.device)
    pb = torch.empty((gy, 2 * C), dtype=torch.float32, device=x.device) if want_bn else None
    if res is not None:  # g = masked dgrad + res (the other gradient of a forked activation)
        assert dil == 1 and tuple(res.shape) == (N, H, W, C) and res.dtype == x.dtype
        LIB.call("seg_dwconv3x3_bwd_fused_add", _DT[x.dtype], _p(dy), lddy, _p(x), ldx, N, H, W,
                 C, _p(w), layout, mode, _p(ps), _p(pt), _p(res), nhwc(res)[4], _p(g), C, _p(pw),
                 _p(pb), gy, _stream())
    else:
        LIB.call("seg_dwconv3x3_bwd_fused", _DT[x.dtype], _p(dy), lddy, _p(x), ldx, N, H, W, C,
                 _p(w), layout, dil, mode, _p(ps), _p(pt), _p(g), C, _p(pw), _p(pb), gy, _stream())
    if raw_dw:  # the caller reduces pw together with pb (dw_bwd_finalize)
        return g, pw, pb
    if torch_layout:
        dW = torch.empty((C, 1, 3, 3), dtype=torch.float32, device=x.device)
        LIB.call("seg_dwconv3x3_wgrad_finalize", _p(pw), gy, C, _p(dW), _stream())
        return g, dW, pb
    return g, colsum(pw, f64=False).view(9, C), pb


def dwconv_bwd_fused_s2(x, dy, w, pro=None, want_bn=False, raw_dw=False):
    """stride-2 (pad 1, dil 1) depthwise backward in one pass over dy and x: returns (g masked by
    the prologue's ReLU, dW fp32 [C,1,3,3], bn_partial fp32 [gy, 2C] | None).  w: the [C,1,3,3]
    parameter."""
    N, H, W, C, ldx = nhwc(x)
    Nd, Ho, Wo, Cd, lddy = nhwc(dy)
    assert (Nd, Ho, Wo, Cd) == (N, (H + 1) // 2, (W + 1) // 2, C) and tuple(w.shape) == (C, 1, 3, 3)
    mode, ps, pt = _pro(pro)
    g = torch.empty((N, H, W, C), dtype=x.dtype, device=x.device)
    gy = LIB.query("seg_dwconv3x3_s2_grid_y", C, N, H, W)
    pw = torch.empty((gy, 9 * C), dtype=torch.float32, device=x.device)
    pb = torch.empty((gy, 2 * C), dtype=torch.float32, device=x.device) if want_bn else None
    LIB.call("seg_dwconv3x3_s2_bwd_fused", _DT[x.dtype], _p(dy), lddy, _p(x), ldx, N, H, W, C,
             _p(w), mode, _p(ps), _p(pt), _p(g), C, _p(pw), _p(pb), gy, _stream())
    if raw_dw:
        return g, pw, pb
    dW = torch.empty((C, 1, 3, 3), dtype=torch.float32, device=x.device)
    LIB.call("seg_dwconv3x3_wgrad_finalize", _p(pw), gy, C, _p(dW), _stream())
    return g, dW, pb


def dwconv_wgrad(x, dy, stride, dil, pro=None, torch_layout=False):
    """-> fp32 [9, C], or with torch_layout the parameter's own [C, 1, 3, 3]."""
    N, Hi, Wi, C, ldx = nhwc(x)
    _, Ho, Wo, _, lddy = nhwc(dy)
    mode, ps, pt = _pro(pro)
    gy = LIB.query("seg_dwconv_grid_y", _DT[x.dtype], C, N, Ho, Wo, stride, dil, 2)
    partial = torch.empty((gy, 9 * C), dtype=torch.float32, device=x.device)
    LIB.call("seg_dwconv3x3_wgrad", _DT[x.dtype], _p(x), ldx, N, Hi, Wi, C, _p(dy), lddy, Ho, Wo,
             stride, dil, mode, _p(ps), _p(pt), _p(partial), gy, _stream())
    if torch_layout:
        out = torch.empty((C, 1, 3, 3), dtype=torch.float32, device=x.device)
        LIB.call("seg_dwconv3x3_wgrad_finalize", _p(partial), gy, C, _p(out), _stream())
        return out
    return colsum(partial, f64=False).view(9, C)


def bn_finalize(sums, count, gamma, beta, eps, momentum, running_mean, running_var,
                mean_offset=None):
    C = sums.numel() // 2
    dev = sums.device
    out = torch.empty((4, C), dtype=torch.float32, device=dev)
    LIB.call("seg_bn_finalize", _p(sums), *_count(count), _p(gamma), _p(beta), float(eps),
             float(momentum), _p(running_mean), _p(running_var), _p(out[0]), _p(out[1]),
             _p(out[2]), _p(out[3]), C, _p(mean_offset), _stream())
    return out[0], out[1], out[2], out[3]  # mean, invstd, scale, shift


def bn_eval_affine(gamma, beta, rm, rv, eps):
    C = rm.numel()
    out = torch.empty((2, C), dtype=torch.float32, device=rm.device)
    LIB.call("seg_bn_eval_affine", _p(gamma), _p(beta), _p(rm), _p(rv), float(eps), _p(out[0]),
             _p(out[1]), C, _stream())
    return out[0], out[1]


def bn_eval_affine_multi(jobs):
    """jobs: [(gamma|None, beta|None, running_mean, running_var, eps)] -> [(scale, shift)] by one
    launch per 48 BatchNorms."""
    import ctypes
    n = len(jobs)
    outs = [torch.empty((2, j[2].numel()), dtype=torch.float32, device=j[2].device) for j in jobs]
    vp, fp, ip = ctypes.c_void_p * n, ctypes.c_float * n, ctypes.c_int * n
    LIB.call("seg_bn_eval_affine_multi", n, vp(*[_p(j[0]) for j in jobs]),
             vp(*[_p(j[1]) for j in jobs]), vp(*[_p(j[2]) for j in jobs]),
             vp(*[_p(j[3]) for j in jobs]), fp(*[float(j[4]) for j in jobs]),
             vp(*[o.data_ptr() for o in outs]), ip(*[j[2].numel() for j in jobs]), _stream())
    return [(o[0], o[1]) for o in outs]


def bn_apply(x, pro_x=None, r=None, pro_r=None, chan_mul=None, post_relu=False, out=None,
             elem_mul=None):
    N, H, W, C, ldx = nhwc(x)
    mx, sx, tx = _pro(pro_x)
    mr, sr, tr = _pro(pro_r)
    if out is None:
        out = torch.empty((N, H, W, C), dtype=x.dtype, device=x.device)
    ldy = nhwc(out)[4]
    ldr = nhwc(r)[4] if r is not None else 0
    ldm = nhwc(elem_mul)[4] if elem_mul is not None else 0
    LIB.call("seg_bn_apply", _DT[x.dtype], _p(x), ldx, mx, _p(sx), _p(tx), _p(r), ldr, mr, _p(sr),
             _p(tr), _p(chan_mul), H * W, _p(elem_mul), ldm, int(post_relu), _p(out), ldy,
             N * H * W, C, _stream())
    return out


def sum_n(tensors):
    """NHWC tensors of one shape / dtype -> their sum (fp32 accumulation in list order, one
    rounding): ONE launch for up to 8 operands (more: in groups)."""
    import ctypes
    ts = list(tensors)
    while len(ts) > 1:
        group, ts = ts[:8], ts[8:]
        N, H, W, C, _ = nhwc(group[0])
        lds = [nhwc(t)[4] for t in group]
        assert all(t.shape == group[0].shape and t.dtype == group[0].dtype for t in group)
        out = torch.empty((N, H, W, C), dtype=group[0].dtype, device=group[0].device)
        n = len(group)
        LIB.call("seg_sum_n", _DT[out.dtype], n, (ctypes.c_void_p * n)(*[t.data_ptr() for t in group]),
                 (ctypes.c_long * n)(*lds), _p(out), C, N * H * W, C, _stream())
        ts.insert(0, out)
    return ts[0]


def nearest_add(x, pro_x, r, pro_r, shift, post_relu=False, out=None):
    """y = post_relu?(act_x(x) + act_r(nearest_upsample_{2^shift}(r)))  (hrnet.py:186,215-229)"""
    N, H, W, C, ldx = nhwc(x)
    Nr, Hr, Wr, Cr, ldr = nhwc(r)
    if (Nr, Hr << shift, Wr << shift, Cr) != (N, H, W, C):
        raise ValueError("nearest_add: %s is not %s upsampled by 2^%d"
                         % (tuple(x.shape), tuple(r.shape), shift))
    mx, sx, tx = _pro(pro_x)
    mr, sr, tr = _pro(pro_r)
    if out is None:
        out = torch.empty((N, H, W, C), dtype=x.dtype, device=x.device)
    LIB.call("seg_nearest_add", _DT[x.dtype], _p(x), ldx, mx, _p(sx), _p(tx), _p(r), ldr, mr,
             _p(sr), _p(tr), shift, int(post_relu), _p(out), nhwc(out)[4], N, H, W, C, _stream())
    return out


def nearest_sum_bwd(g, shift):
    """gradient of nearest_upsample_{2^shift}: 2^shift x 2^shift block sums."""
    N, H, W, C, ldg = nhwc(g)
    out = torch.empty((N, H >> shift, W >> shift, C), dtype=g.dtype, device=g.device)
    LIB.call("seg_nearest_sum_bwd", _DT[g.dtype], _p(g), ldg, N, H, W, C, shift, _p(out), C,
             _stream())
    return out


def bn_bwd_reduce_partial(g, x, pro, chan_mul=None, elem_mul=None):
    """-> fp32 [grid_y, 2C] per-block (sum g', sum g'*x)."""
    N, H, W, C, ldg = nhwc(g)
    ldx = nhwc(x)[4]
    mode, s, t = _pro(pro)
    M = N * H * W
    gy = LIB.query("seg_bn_bwd_grid_y", _DT[g.dtype], C, M)
    partial = torch.empty((gy, 2 * C), dtype=torch.float32, device=g.device)
    ldm = nhwc(elem_mul)[4] if elem_mul is not None else 0
    LIB.call("seg_bn_bwd_reduce", _DT[g.dtype], _p(g), ldg, _p(x), ldx, mode, _p(s), _p(t),
             _p(chan_mul), H * W, _p(elem_mul), ldm, M, C, _p(partial), gy, _stream())
    return partial


def bn_bwd_small(g, x, pro, count, mean, invstd, gamma, chan_mul=None, elem_mul=None,
                 training=True, out=None):
    """BatchNorm backward of a small tensor (<= SMALL_BN_ROWS samples per channel) in one
    float64 launch -> (dx, dgamma, dbeta)."""
    N, H, W, C, ldg = nhwc(g)
    ldx = nhwc(x)[4]
    mode, s, t = _pro(pro)
    if out is None:
        out = torch.empty((N, H, W, C), dtype=g.dtype, device=g.device)
    lddx = nhwc(out)[4]
    dgb = torch.empty((2, C), dtype=torch.float32, device=g.device)
    ldm = nhwc(elem_mul)[4] if elem_mul is not None else 0
    LIB.call("seg_bn_bwd_small", _DT[g.dtype], _p(g), ldg, _p(x), ldx, _p(out), lddx, N * H * W, C,
             mode, _p(s), _p(t), _p(mean), _p(invstd), _p(gamma), _p(chan_mul), H * W,
             _p(elem_mul), ldm, float(count), 1 if training else 0, _p(dgb[0]), _p(dgb[1]),
             _stream())
    return out, dgb[0], dgb[1]


def bn_bwd_reduce(g, x, pro, chan_mul=None):
    """-> float64 [2C]: (sum g', sum g'*x)."""
    return colsum(bn_bwd_reduce_partial(g, x, pro, chan_mul), f64=True)


def _ws(R, C, dev):
    return torch.empty(128 * C, dtype=torch.float64, device=dev) if R > 1024 else None


def bn_finalize_p(partial, count, gamma, beta, eps, momentum, running_mean, running_var,
                  mean_offset=None):
    """partial fp32 [R, 2, C] (or [R, 2C]) -> mean, invstd, scale, shift (one fused launch)."""
    R = partial.shape[0]
    C = partial.numel() // (2 * R)
    out = torch.empty((4, C), dtype=torch.float32, device=partial.device)
    ws = _ws(R, C, partial.device)
    LIB.call("seg_bn_finalize_p", _p(partial), R, float(count), _p(gamma), _p(beta), float(eps),
             float(momentum), _p(running_mean), _p(running_var), _p(out[0]), _p(out[1]),
             _p(out[2]), _p(out[3]), C, _p(mean_offset), _p(ws), _stream())
    return out[0], out[1], out[2], out[3]


SMALL_BN_ROWS = 1024


def bn_finalize_small(y, gamma, beta, eps, momentum, running_mean, running_var, mean_offset=None):
    """BatchNorm statistics of a SMALL stored tensor y [N,H,W,C] (N*H*W <= 4096 rows), two-pass
    (mean, then sum (x - mean)^2) -> mean, invstd, scale, shift (+ running-stat update)."""
    N, H, W, C, ldy = nhwc(y)
    out = torch.empty((4, C), dtype=torch.float32, device=y.device)
    LIB.call("seg_bn_finalize_small", _DT[y.dtype], _p(y), ldy, N * H * W, C, _p(gamma), _p(beta),
             float(eps), float(momentum), _p(running_mean), _p(running_var), _p(out[0]),
             _p(out[1]), _p(out[2]), _p(out[3]), _p(mean_offset), _stream())
    return out[0], out[1], out[2], out[3]


def bn_moments_small(y):
    """This rank's two-pass moments of a small stored tensor as float64 sums [2C + 1] =
    (n*mean | M2 + n*mean^2 | n): the SyncBatchNorm forward message of a small BatchNorm."""
    N, H, W, C, ldy = nhwc(y)
    out = torch.empty(2 * C + 1, dtype=torch.float64, device=y.device)
    LIB.call("seg_bn_moments_small", _DT[y.dtype], _p(y), ldy, N * H * W, C, _p(out), _stream())
    return out


def bn_finalize_small_sync(box, y, gamma, beta, eps, momentum, running_mean, running_var,
                           mean_offset=None):
    """bn_finalize_small for SyncBatchNorm through the peer mailbox (one launch) -> mean, invstd,
    scale, shift, global count (float64 [1])."""
    N, H, W, C, ldy = nhwc(y)
    out = torch.empty((4, C), dtype=torch.float32, device=y.device)
    cnt = torch.empty(1, dtype=torch.float64, device=y.device)
    LIB.call("seg_bn_finalize_small_sync", box.handle, _DT[y.dtype], _p(y), ldy, N * H * W, C,
             _p(gamma), _p(beta), float(eps), float(momentum), _p(running_mean), _p(running_var),
             _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]), _p(mean_offset), _p(cnt), _stream())
    return out[0], out[1], out[2], out[3], cnt


def bn_finalize_p_sync(box, partial, local_count, gamma, beta, eps, momentum, running_mean,
                       running_var, mean_offset=None):
    """bn_finalize_p for SyncBatchNorm in ONE launch: this rank's partial rows are summed, the
    sums and the local element count exchanged through the peer mailbox inside the kernel, and
    the global statistics finalized -> mean, invstd, scale, shift, global count (float64 [1])."""
    R = partial.shape[0]
    C = partial.numel() // (2 * R)
    out = torch.empty((4, C), dtype=torch.float32, device=partial.device)
    cnt = torch.empty(1, dtype=torch.float64, device=partial.device)
    ws = _ws(R, C, partial.device)
    LIB.call("seg_bn_finalize_p_sync", box.handle, _p(partial), R, float(local_count), _p(gamma),
             _p(beta), float(eps), float(momentum), _p(running_mean), _p(running_var), _p(out[0]),
             _p(out[1]), _p(out[2]), _p(out[3]), C, _p(mean_offset), _p(cnt), _p(ws), _stream())
    return out[0], out[1], out[2], out[3], cnt


def bn_bwd_finalize_p_sync(box, partial, count_dev, mean, invstd, gamma, grad_scale):
    C = mean.numel()
    out = torch.empty((4, C), dtype=torch.float32, device=mean.device)
    ws = _ws(partial.shape[0], C, mean.device)
    LIB.call("seg_bn_bwd_finalize_p_sync", box.handle, _p(partial), partial.shape[0], _p(count_dev),
             _p(mean), _p(invstd), _p(gamma), _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]), C,
             float(grad_scale), _p(ws), _stream())
    return out[0], out[1], out[2], out[3]


def bn_bwd_finalize_p(partial, count, mean, invstd, gamma):
    R = partial.shape[0]
    C = mean.numel()
    out = torch.empty((4, C), dtype=torch.float32, device=mean.device)
    ws = _ws(R, C, mean.device)
    LIB.call("seg_bn_bwd_finalize_p", _p(partial), R, float(count), _p(mean), _p(invstd),
             _p(gamma), _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]), C, _p(ws), _stream())
    return out[0], out[1], out[2], out[3]


def bn_bwd_finalize(sums, count, mean, invstd, gamma, grad_scale=1.0):
    """grad_scale multiplies dgamma / dbeta only (SyncBatchNorm: 1 / world size)."""
    C = mean.numel()
    out = torch.empty((4, C), dtype=torch.float32, device=mean.device)
    LIB.call("seg_bn_bwd_finalize_s", _p(sums), *_count(count), _p(mean), _p(invstd), _p(gamma),
             _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]), C, float(grad_scale), _stream())
    return out[0], out[1], out[2], out[3]  # dgamma, dbeta, c0, c1


def bn_bwd_apply(g, x, pro, c0=None, c1=None, chan_mul=None, out=None, elem_mul=None):
    N, H, W, C, ldg = nhwc(g)
    ldx = nhwc(x)[4]
    mode, s, t = _pro(pro)
    if out is None:
        out = torch.empty((N, H, W, C), dtype=g.dtype, device=g.device)
    lddx = nhwc(out)[4]
    ldm = nhwc(elem_mul)[4] if elem_mul is not None else 0
    LIB.call("seg_bn_bwd_apply", _DT[g.dtype], _p(g), ldg, _p(x), ldx, mode, _p(s), _p(t), _p(c0),
             _p(c1), _p(chan_mul), H * W, _p(elem_mul), ldm, _p(out), lddx, N * H * W, C,
             _stream())
    return out


# ----------------------------------------------------------------------------- BN fold
def fold_weights(w2d, scale, shift, dtype, want_transpose=False, want_bias=True):
    """fp32 W [O,C] -> (W*scale in dtype, its transpose [C,O] or None, W@shift fp32 or None)."""
    O, C = w2d.shape
    dev = w2d.device
    wp = torch.empty((O, C), dtype=dtype, device=dev)
    wpt = torch.empty((C, O), dtype=dtype, device=dev) if want_transpose else None
    bp = torch.empty(O, dtype=torch.float32, device=dev) if want_bias else None
    LIB.call("seg_fold_weights", _DT[dtype], _p(w2d), _p(scale), _p(shift), _p(wp), _p(wpt),
             _p(bp), O, C, _stream())
    return wp, wpt, bp


def fold_bwd_reduce(w2d, dwp, scale, shift, db=None):
    """dwp: dW' [O, C] or the weight-gradient split partials [S, O*C].
    -> (dW fp32 [O,C], dsdt fp32 [R, 2C] partial rows for fold_bwd_finalize)."""
    O, C = w2d.shape
    S = 1 if tuple(dwp.shape) == (O, C) else dwp.shape[0]
    assert dwp.numel() == S * O * C and dwp.is_contiguous()
    R = LIB.query("seg_fold_bwd_rows", O)
    if S > 8 and R == 1:
        # a narrow conv on a huge map (decoder c1_block: 256 -> 48 on 263 k pixels, 256 pixel
        # splits): the fold reduction would walk all S partials with ONE row group of blocks
        # (184 us for 12.6 MB) — sum the splits with the wide column-sum kernel first
        dwp = colsum(dwp.view(S, O * C), f64=False).view(O, C)
        S = 1
    dW = torch.empty((O, C), dtype=torch.float32, device=w2d.device)
    dsdt = torch.empty((R, 2 * C), dtype=torch.float32, device=w2d.device)
    LIB.call("seg_fold_bwd_reduce", _p(w2d), _p(dwp), S, _p(scale), _p(shift), _p(db), _p(dW),
             _p(dsdt), O, C, _stream())
    return dW, dsdt


def fold_bwd_finalize(dsdt, count, mean, invstd, gamma, scale, grad_scale=1.0):
    C = mean.numel()
    out = torch.empty((4, C), dtype=torch.float32, device=mean.device)
    dsdt = dsdt.view(-1, 2 * C)
    LIB.call("seg_fold_bwd_finalize_s", _p(dsdt), dsdt.shape[0], *_count(count), _p(mean),
             _p(invstd), _p(gamma), _p(scale), _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]), C,
             float(grad_scale), _stream())
    return out[0], out[1], out[2], out[3]  # dgamma, dbeta, c0, c1


def fold_bwd_finalize_sync(box, dsdt, count_dev, mean, invstd, gamma, scale, grad_scale):
    """fold_bwd_finalize for SyncBatchNorm: dsdt = this rank's LOCAL (ds, dt) partial rows; the
    exchange happens inside the kernel."""
    C = mean.numel()
    out = torch.empty((4, C), dtype=torch.float32, device=mean.device)
    dsdt = dsdt.view(-1, 2 * C)
    LIB.call("seg_fold_bwd_finalize_sync", box.handle, _p(dsdt), dsdt.shape[0], _p(count_dev),
             _p(mean), _p(invstd), _p(gamma), _p(scale), _p(out[0]), _p(out[1]), _p(out[2]),
             _p(out[3]), C, float(grad_scale), _stream())
    return out[0], out[1], out[2], out[3]


# ----------------------------------------------------------------------------- pooling
def maxpool(x, k, stride, pad, pro=None):
    """-> (y, idx uint8): y = maxpool(act(x)), -inf padding."""
    N, Hi, Wi, C, ldx = nhwc(x)
    Ho, Wo = (Hi + 2 * pad - k) // stride + 1, (Wi + 2 * pad - k) // stride + 1
    mode, ps, pt = _pro(pro)
    y = torch.empty((N, Ho, Wo, C), dtype=x.dtype, device=x.device)
    idx = torch.empty((N, Ho, Wo, C), dtype=torch.uint8, device=x.device)
    LIB.call("seg_maxpool_fwd", _DT[x.dtype], _p(x), ldx, N, Hi, Wi, C, k, stride, pad, mode,
             _p(ps), _p(pt), _p(y), C, Ho, Wo, _p(idx), _stream())
    return y, idx


def maxpool_bwd(gy, idx, in_hw, k, stride, pad):
    N, Ho, Wo, C, ldgy = nhwc(gy)
    Hi, Wi = in_hw
    gx = torch.empty((N, Hi, Wi, C), dtype=gy.dtype, device=gy.device)
    LIB.call("seg_maxpool_bwd", _DT[gy.dtype], _p(gx), C, N, Hi, Wi, C, k, stride, pad, _p(gy),
             ldgy, Ho, Wo, _p(idx), _stream())
    return gx


def adaptive_avgpool_sums(x, o):
    """-> fp32 [N, o, o, C] bin SUMS (caller divides by the bin areas)."""
    N, H, W, C, ldx = nhwc(x)
    chunks = LIB.query("seg_adaptive_avgpool_chunks", H, W, o)
    partial = torch.empty((chunks, N * o * o * C), dtype=torch.float32, device=x.device)
    LIB.call("seg_adaptive_avgpool_partial", _DT[x.dtype], _p(x), ldx, N, H, W, C, o, _p(partial),
             chunks, _stream())
    sums = partial[0] if chunks == 1 else colsum(partial, f64=False)
    return sums.view(N, o, o, C)


def adaptive_avgpool_bwd(gy, in_hw):
    N, o, _, C, ldgy = nhwc(gy)
    H, W = in_hw
    gx = torch.empty((N, H, W, C), dtype=gy.dtype, device=gy.device)
    LIB.call("seg_adaptive_avgpool_bwd", _DT[gy.dtype], _p(gx), C, N, H, W, C, o, _p(gy), ldgy,
             _stream())
    return gx


_BIN_AREAS = {}


def adaptive_bin_areas(H, W, o, device):
    """[o, o] fp32 areas of ATen's adaptive-pool bins (cached per geometry: a host-to-device
    copy must not happen inside a HIP-graph capture)."""
    key = (H, W, o, str(device))
    if key not in _BIN_AREAS:
        _BIN_AREAS[key] = _adaptive_bin_areas(H, W, o, device)
    return _BIN_AREAS[key]


def _adaptive_bin_areas(H, W, o, device):
    import math
    hs = [math.ceil((i + 1) * H / o) - (i * H) // o for i in range(o)]
    ws = [math.ceil((j + 1) * W / o) - (j * W) // o for j in range(o)]
    return torch.tensor([[float(a * b) for b in ws] for a in hs], device=device)


# ----------------------------------------------------------------------------- resize
def bilinear(x, out_hw, pro=None, chan_mul=None, align_corners=True, out=None):
    N, Hi, Wi, C, ldx = nhwc(x)
    Ho, Wo = out_hw
    mode, s, t = _pro(pro)
    if out is None:
        out = torch.empty((N, Ho, Wo, C), dtype=x.dtype, device=x.device)
    ldy = nhwc(out)[4]
    LIB.call("seg_bilinear_fwd", _DT[x.dtype], _p(x), ldx, N, Hi, Wi, C, mode, _p(s), _p(t),
             _p(chan_mul), _p(out), ldy, Ho, Wo, int(align_corners), _stream())
    return out


def bilinear_bwd(gy, in_hw, align_corners=True):
    N, Ho, Wo, C, ldgy = nhwc(gy)
    Hi, Wi = in_hw
    gx = torch.empty((N, Hi, Wi, C), dtype=gy.dtype, device=gy.device)
    LIB.call("seg_bilinear_bwd", _DT[gy.dtype], _p(gx), C, N, Hi, Wi, C, _p(gy), ldgy, Ho, Wo,
             int(align_corners), _stream())
    return gx


def upsample_to_nchw(x, C, out_hw, align_corners=True):
    """x: NHWC view whose first C channels are valid -> float32 NCHW [N,C,Ho,Wo]."""
    N, Hi, Wi, _, ldx = nhwc(x)
    Ho, Wo = out_hw
    out = torch.empty((N, C, Ho, Wo), dtype=torch.float32, device=x.device)
    LIB.call("seg_upsample_to_nchw", _DT[x.dtype], _p(x), ldx, N, Hi, Wi, C, _p(out), Ho, Wo,
             int(align_corners), _stream())
    return out


def upsample_to_nchw_bwd(gy, in_hw, dtype, pitch, align_corners=True):
    """gy float32 NCHW -> NHWC [N,Hi,Wi,pitch] of `dtype` (channels >= C are zero)."""
    if not gy.is_cuda or gy.dtype != torch.float32:
        raise RuntimeError("upsample_to_nchw_bwd needs a float32 HIP device tensor")
    N, C, Ho, Wo = gy.shape
    Hi, Wi = in_hw
    gy = gy.contiguous()
    gx = torch.empty((N, Hi, Wi, pitch), dtype=dtype, device=gy.device)
    LIB.call("seg_upsample_to_nchw_bwd", _DT[dtype], _p(gx), pitch, N, Hi, Wi, C, _p(gy), Ho, Wo,
             int(align_corners), _stream())
    return gx


def nchw_to_nhwc_pad(x, dtype):
    """float32 NCHW image (C <= 16B/elem) -> NHWC [N,H,W,VEC] zero-padded."""
    if not x.is_cuda:
        raise RuntimeError("segmentron_amd ops need HIP device tensors (no CPU fallback)")
    N, Cin, H, W = x.shape
    x = x.contiguous().float()
    y = torch.empty((N, H, W, vec_of(dtype)), dtype=dtype, device=x.device)
    LIB.call("seg_nchw_to_nhwc_pad", _DT[dtype], _p(x), N, Cin, H, W, _p(y), _stream())
    return y


# ----------------------------------------------------------------------------- fused loss tail
def upsample_ce_fwd(lo, target, out_hw, ignore_index, align_corners=True):
    """lo: NHWC logits view [N,Hi,Wi,C] of a channel-padded buffer; target int64 [N,H,W].
    -> float32[2] device tensor (mean cross-entropy over valid pixels, 1 / valid count)."""
    N, Hi, Wi, C, ld = nhwc(lo)
    H, W = out_hw
    if not target.is_cuda or target.dtype != torch.int64 or tuple(target.shape) != (N, H, W):
        raise RuntimeError("upsample_ce: target must be an int64 HIP tensor [N, H, W] = %s, got %s"
                           % ((N, H, W), tuple(target.shape)))
    target = target.contiguous()
    blocks = LIB.query("seg_upsample_ce_blocks", N, H, W)
    ws = torch.empty(2 * blocks, dtype=torch.float64, device=lo.device)
    out = torch.empty(2, dtype=torch.float32, device=lo.device)
    LIB.call("seg_upsample_ce_fwd", _DT[lo.dtype], _p(lo), ld, N, Hi, Wi, C, _p(target), H, W,
             int(ignore_index), int(align_corners), _p(ws), _p(out), _stream())
    return out


def upsample_ce_bwd(lo, target, out_hw, ignore_index, loss_out, grad_out, pitch,
                    align_corners=True):
    """-> d(loss)/d(lo) * grad_out as NHWC [N,Hi,Wi,pitch] in lo.dtype (channels >= C zero)."""
    N, Hi, Wi, C, ld = nhwc(lo)
    H, W = out_hw
    grad_out = grad_out.reshape(1).to(torch.float32).contiguous()
    dlo = torch.empty((N, Hi, Wi, pitch), dtype=lo.dtype, device=lo.device)
    LIB.call("seg_upsample_ce_bwd", _DT[lo.dtype], _p(lo), ld, N, Hi, Wi, C,
             _p(target.contiguous()), H, W, int(ignore_index), int(align_corners), _p(loss_out),
             _p(grad_out), _p(dlo), pitch, _stream())
    return dlo


# ----------------------------------------------------------------------------- optimizer
def sgd_check(p, g, b):
    if not (p.is_cuda and g.is_cuda and b.is_cuda):
        raise RuntimeError("segmentron_amd SGD needs HIP device tensors (no CPU fallback)")
    if not (p.dtype == g.dtype == b.dtype == torch.float32):
        raise RuntimeError("segmentron_amd SGD: parameters / gradients / buffers are fp32")
    if not (p.is_contiguous() and g.is_contiguous() and b.is_contiguous()):
        raise RuntimeError("segmentron_amd SGD: non-contiguous tensor")
    if g.numel() != p.numel() or b.numel() != p.numel():
        raise RuntimeError("segmentron_amd SGD: size mismatch")


def pack_multi(jobs, dtype):
    """jobs: [(src fp32 [O, C] contiguous, transpose)] -> list of packed tensors in `dtype`
    ([O, C] or [C, O]) written by as few seg_pack_multi launches as possible."""
    import ctypes
    n = len(jobs)
    outs = []
    for src, tr in jobs:
        assert src.dtype == torch.float32 and src.is_cuda and src.is_contiguous() and src.dim() == 2
        O, C = src.shape
        outs.append(torch.empty((C, O) if tr else (O, C), dtype=dtype, device=src.device))
    vp = ctypes.c_void_p * n
    ci = ctypes.c_int * n
    LIB.call("seg_pack_multi", _DT[dtype], n, vp(*[j[0].data_ptr() for j in jobs]),
             vp(*[o.data_ptr() for o in outs]), ci(*[j[0].shape[0] for j in jobs]),
             ci(*[j[0].shape[1] for j in jobs]), ci(*[1 if j[1] else 0 for j in jobs]), _stream())
    return outs


def sgd_plan(params, bufs, groups):
    """The per-step-invariant half of a multi-tensor SGD launch: ctypes arrays of the parameter /
    buffer pointers, sizes and group indices (440 tensors: rebuilt only when they change)."""
    import ctypes
    n = len(params)
    vp = ctypes.c_void_p * n
    return (n, vp, vp(*[p.data_ptr() for p in params]), vp(*[b.data_ptr() for b in bufs]),
            (ctypes.c_long * n)(*[p.numel() for p in params]), (ctypes.c_int * n)(*groups))


def sgd_multi_tensor(plan, grads, lr_dev, wd_dev, momentum, first):
    """One fused SGD(momentum, weight decay) step over the tensors of `plan` (sgd_plan) with
    this step's gradient tensors (csrc/optim.hip); the plan's group indices address the DEVICE
    float arrays lr_dev / wd_dev."""
    n, vp, pp, pb, pn, pg = plan
    LIB.call("seg_sgd_multi_tensor", n, pp, vp(*[g.data_ptr() for g in grads]), pb, pn, pg,
             _p(lr_dev), _p(wd_dev), float(momentum), int(bool(first)), _stream())


# ----------------------------------------------------------------------------- metrics
def metric_counters(nclass, device):
    """Zeroed int64 [2 + 3*nclass]: correct, labelled, inter[], pred[], lab[] (csrc/metric.hip)."""
    return torch.zeros(2 + 3 * nclass, dtype=torch.int64, device=device)


def metric_update_nchw(logits, target, nclass, counters):
    """logits fp32 NCHW, target int64 [N,H,W]: accumulates score.py:83-113's counts."""
    if not (logits.is_cuda and target.is_cuda and counters.is_cuda):
        raise RuntimeError("segmentron_amd metrics need HIP device tensors (no CPU fallback)")
    if logits.dtype != torch.float32 or not logits.is_contiguous():
        logits = logits.float().contiguous()
    if target.dtype != torch.int64 or not target.is_contiguous():
        target = target.long().contiguous()
    N, C, H, W = logits.shape
    assert tuple(target.shape) == (N, H, W) and counters.numel() == 2 + 3 * nclass
    LIB.call("seg_metric_update_nchw", _p(logits), N, C, H, W, _p(target), nclass, _p(counters),
             _stream())
    return counters


def metric_update_upsample(lo, target, align_corners, nclass, counters):
    """lo: NHWC logits [N,Hi,Wi,C]; the counts are taken on their bilinear upsample to the
    target's [H, W] without materialising it."""
    if not (lo.is_cuda and target.is_cuda and counters.is_cuda):
        raise RuntimeError("segmentron_amd metrics need HIP device tensors (no CPU fallback)")
    N, Hi, Wi, C, ld = nhwc(lo)
    if target.dtype != torch.int64 or not target.is_contiguous():
        target = target.long().contiguous()
    assert target.shape[0] == N and counters.numel() == 2 + 3 * nclass
    H, W = target.shape[1:]
    LIB.call("seg_metric_update_upsample", _DT[lo.dtype], _p(lo), ld, N, Hi, Wi, C, _p(target), H,
             W, int(bool(align_corners)), nclass, _p(counters), _stream())
    return counters


# ----------------------------------------------------------------------------- DANet attention
def row_softmax(e, L, out_dtype, sign=1.0):
    """e [R, Lp] (fp32 or bf16, row-major) -> a [R, Lp] in out_dtype: softmax over the first L
    columns of sign * e, zeros in the padding columns."""
    R, Lp = e.shape
    assert e.is_cuda and e.stride(1) == 1
    a = torch.empty((R, Lp), dtype=out_dtype, device=e.device)
    LIB.call("seg_row_softmax", _p(e), _DT[e.dtype], e.stride(0), _p(a), _DT[out_dtype], Lp, R, L,
             Lp, float(sign), _stream())
    return a


def row_softmax_bwd(a, g, L, out_dtype, sign=1.0):
    """de = sign * a * (g - sum_j a g) over the first L columns (zeros behind)."""
    R, Lp = a.shape
    assert a.stride(1) == 1 and g.stride(1) == 1 and tuple(g.shape) == (R, Lp)
    de = torch.empty((R, Lp), dtype=out_dtype, device=a.device)
    LIB.call("seg_row_softmax_bwd", _p(a), _DT[a.dtype], a.stride(0), _p(g), _DT[g.dtype],
             g.stride(0), _p(de), _DT[out_dtype], Lp, R, L, Lp, float(sign), _stream())
    return de


def gemm_nt(a, b):
    """a [P, K], b [O, K] (both compute dtype, row-major, K and O multiples of the channel
    vector) -> a @ b.T [P, O] in the compute dtype: a 1x1 convolution of the P "pixels" of a
    (seg_conv_gemm_fwd — fp32 MFMA accumulation)."""
    P, Kd = a.shape
    y, _ = conv_gemm(a.view(1, 1, P, Kd), b, b.shape[0], 1, 1, 1, 0, 1)
    return y.view(P, b.shape[0])


def gemm_tn(a, b):
    """a [P, M], b [P, K] -> a.T @ b [M, K] float32: the weight-gradient GEMM of a 1x1
    convolution (seg_conv_gemm_wgrad: sum over the P pixels, fixed-order fp32 partials)."""
    P, M = a.shape
    return conv_wgrad(b.view(1, 1, P, b.shape[1]), a.view(1, 1, P, M), M, 1, 1, 1, 0, 1)


# ----------------------------------------------------------------------------- criss-cross attention
def cca_attention(q, k):
    """q, k NHWC [N,H,W,C'] -> fp32 attention [N,H,W,W+H-1] = softmax over the criss-cross set."""
    N, H, W, C, ldq = nhwc(q)
    ldk = nhwc(k)[4]
    att = torch.empty((N, H, W, H + W - 1), dtype=torch.float32, device=q.device)
    LIB.call("seg_cca_attention", _DT[q.dtype], _p(q), ldq, _p(k), ldk, N, H, W, C, _p(att),
             _stream())
    return att


def cca_attention_bwd(dout, v, att, scale=None):
    """dE (fp32, like att): softmax backward of dA = scale * dout . v[partner]."""
    N, H, W, C, lddo = nhwc(dout)
    ldv = nhwc(v)[4]
    de = torch.empty_like(att)
    LIB.call("seg_cca_attention_bwd", _DT[v.dtype], _p(dout), lddo, _p(v), ldv, N, H, W, C,
             _p(att), _p(scale), _p(de), _stream())
    return de


def cca_map(wt, b, transposed=False, gamma=None, res=None, want_raw=False):
    """out[p] = gamma * sum_z wt(p, z) * b[partner(p, z)] (+ res[p]); transposed: summed over the
    pixels attending TO p.  -> out, or (out, raw sum) with want_raw."""
    N, H, W, C, ldb = nhwc(b)
    out = torch.empty((N, H, W, C), dtype=b.dtype, device=b.device)
    raw = torch.empty_like(out) if want_raw else None
    ldres = nhwc(res)[4] if res is not None else 0
    LIB.call("seg_cca_map", _DT[b.dtype], _p(wt), _p(b), ldb, N, H, W, C, int(bool(transposed)),
             _p(gamma), _p(res), ldres, _p(out), C, _p(raw), C, _stream())
    return (out, raw) if want_raw else out
