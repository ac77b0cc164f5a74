"""Helpers for the GPU parity tests: NCHW CPU float <-> NHWC device tensors, error metrics."""
import torch

DEV = "cuda"


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def quant(x, dtype):
    """Round a CPU fp32 tensor to what `dtype` can hold (identity for fp32)."""
    return x.to(dtype).float()


def to_dev_nhwc(x_nchw, dtype, pitch=None, off=0):
    """CPU NCHW float -> device NHWC [N,H,W,C] (optionally a channel slice of a wider buffer
    filled with NaN to catch out-of-slice reads)."""
    n, c, h, w = x_nchw.shape
    t = x_nchw.permute(0, 2, 3, 1).contiguous().to(dtype)
    if pitch is None:
        return t.to(DEV)
    buf = torch.full((n, h, w, pitch), float("nan"), dtype=dtype)
    buf[..., off:off + c] = t
    return buf.to(DEV)[..., off:off + c]


def to_cpu_nchw(t_nhwc):
    return t_nhwc.detach().float().cpu().permute(0, 3, 1, 2).contiguous()


def tol(dtype):
    """(rtol on max-normalised error, elementwise bf16 ulp factor)"""
    return 2e-5 if dtype == torch.float32 else 6e-3


def assert_close(got, ref, dtype, what="", scale=None, fac=1.0):
    got, ref = got.double(), ref.double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), what + ": non-finite values"
    s = ref.abs().max().item() if scale is None else scale
    s = max(s, 1e-12)
    err = (got - ref).abs().max().item() / s
    assert err <= tol(dtype) * fac, "%s: max-normalised error %.3e > %.3e" % (what, err,
                                                                                tol(dtype) * fac)
    return err


def tie_tolerant_argmax_check(got, ref, what):
    """`argmax masks identical` up to the oracle's own exact/near ties: a pixel may differ only
    where the oracle's top-2 margin is within the observed numerical difference (two equally
    valid fp32 evaluation orders of the same graph cannot agree there either)."""
    err = (got - ref).abs().max().item()
    # the tolerance below is derived from the observed error, so the error itself is bounded
    # first (north_star: logits within 1e-3 relative): a large error cannot widen what counts
    # as a tie
    assert err <= 1e-3 * ref.abs().max().item(), "%s: max-abs-diff %.3e above the 1e-3 bar" % (what, err)
    a, b = got.argmax(1), ref.argmax(1)
    diff = a != b
    n_diff = int(diff.sum())
    if n_diff:
        top2 = ref.topk(2, dim=1).values
        margin = (top2[:, 0] - top2[:, 1])[diff]
        assert (margin <= 4 * err).all(), (
            "%s: %d argmax mismatches with oracle margin up to %.3e > 4 x max-abs-diff %.3e"
            % (what, n_diff, margin.max().item(), err))
    assert n_diff <= 1e-4 * a.numel(), "%s: %d near-tie pixels" % (what, n_diff)
    return n_diff
