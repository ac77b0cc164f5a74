"""GPU: torch.ops.segmentron_hip.* (segmentron_amd/torch_ops.py) — forward and autograd of the
registered custom operators against the torch CPU reference of the call they replace
(float64), and torch.library.opcheck (schema, fake-tensor consistency, autograd registration)."""
import pytest
import torch
import torch.nn.functional as TF

import segmentron_amd  # noqa: F401  (registers the operators)
from _util import DEV, assert_close, quant, rnd, to_cpu_nchw, to_dev_nhwc

pytestmark = pytest.mark.gpu
DTYPES = [torch.float32, torch.bfloat16]
IDS = ["fp32", "bf16"]
NS = torch.ops.segmentron_hip


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("case", [(2, 17, 19, 72, 40, 1, 1, 0, 1, True, False),
                                  (2, 33, 65, 728, 728, 1, 1, 0, 1, False, False),
                                  (1, 21, 23, 32, 64, 3, 2, 1, 1, True, True),
                                  (1, 20, 24, 16, 24, 3, 1, 2, 2, False, True),
                                  (2, 13, 15, 256, 19, 1, 1, 0, 1, True, True)])
def test_conv2d_op_forward_backward(case, dtype):
    N, H, W, C, O, k, stride, pad, dil, relu_in, bias = case
    x = quant(rnd((N, C, H, W), 1), dtype)
    w = rnd((O, C, k, k), 2, (2.0 / (C * k * k)) ** 0.5)
    wq = quant(w, dtype)   # the kernel multiplies dtype-rounded weights
    b = rnd((O,), 3, 0.5) if bias else None
    xr = x.double().requires_grad_()
    wr = wq.double().requires_grad_()
    br = b.double().requires_grad_() if bias else None
    xa = torch.relu(xr) if relu_in else xr
    ref = TF.conv2d(xa, wr, br, stride, pad, dil)
    dy = quant(rnd(tuple(ref.shape), 4), dtype)
    ref.backward(dy.double())
    xd = to_dev_nhwc(x, dtype).requires_grad_()
    wd = w.to(DEV).requires_grad_()
    bd = b.to(DEV).requires_grad_() if bias else None
    y = NS.conv2d(xd, wd, bd, stride, pad, dil, relu_in)
    assert_close(to_cpu_nchw(y), ref.detach(), dtype, "conv2d op fwd", fac=2)
    y.backward(to_dev_nhwc(dy, dtype))
    assert_close(to_cpu_nchw(xd.grad), xr.grad, dtype, "conv2d op dx", fac=3)
    assert_close(wd.grad.cpu(), wr.grad, torch.float32, "conv2d op dW",
                 fac=30 if dtype == torch.float32 else 400)
    if bias:
        assert_close(bd.grad.cpu(), br.grad, torch.float32, "conv2d op db", fac=30)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("case", [(2, 17, 19, 128, 1, 1, True), (1, 21, 25, 64, 2, 1, False),
                                  (1, 30, 34, 256, 1, 12, False), (2, 19, 17, 72, 1, 2, True)])
def test_depthwise_conv3x3_op_forward_backward(case, dtype):
    N, H, W, C, stride, dil, relu_in = case
    x = quant(rnd((N, C, H, W), 1), dtype)
    w = rnd((C, 1, 3, 3), 2, 0.4)
    xr, wr = x.double().requires_grad_(), w.double().requires_grad_()
    ref = TF.conv2d(torch.relu(xr) if relu_in else xr, wr, None, stride, dil, dil, groups=C)
    dy = quant(rnd(tuple(ref.shape), 3), dtype)
    ref.backward(dy.double())
    xd = to_dev_nhwc(x, dtype).requires_grad_()
    wd = w.to(DEV).requires_grad_()
    y = NS.depthwise_conv3x3(xd, wd, stride, dil, relu_in)
    assert_close(to_cpu_nchw(y), ref.detach(), dtype, "dw op fwd")
    y.backward(to_dev_nhwc(dy, dtype))
    assert_close(to_cpu_nchw(xd.grad), xr.grad, dtype, "dw op dx")
    assert_close(wd.grad.cpu(), wr.grad, torch.float32, "dw op dW",
                 fac=20 if dtype == torch.float32 else 100)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("align", [True, False])
def test_interpolate_bilinear_op_forward_backward(align, dtype):
    N, C, H, W, Ho, Wo = 2, 48, 9, 13, 33, 49
    x = quant(rnd((N, C, H, W), 1), dtype)
    xr = x.double().requires_grad_()
    ref = TF.interpolate(xr, (Ho, Wo), mode="bilinear", align_corners=align)
    dy = quant(rnd(tuple(ref.shape), 2), dtype)
    ref.backward(dy.double())
    xd = to_dev_nhwc(x, dtype).requires_grad_()
    y = NS.interpolate_bilinear(xd, Ho, Wo, align)
    assert_close(to_cpu_nchw(y), ref.detach(), dtype, "bilinear op fwd")
    y.backward(to_dev_nhwc(dy, dtype))
    assert_close(to_cpu_nchw(xd.grad), xr.grad, dtype, "bilinear op bwd", fac=2)


def test_opcheck_schema_fake_and_autograd_registration():
    from torch.library import opcheck
    utils = ("test_schema", "test_faketensor", "test_autograd_registration")
    x = to_dev_nhwc(rnd((1, 16, 9, 11), 1), torch.float32).requires_grad_()
    w = rnd((24, 16, 3, 3), 2, 0.2).to(DEV).requires_grad_()
    opcheck(NS.conv2d.default, (x, w, None, 1, 1, 1, True), test_utils=utils)
    wd = rnd((16, 1, 3, 3), 3, 0.4).to(DEV).requires_grad_()
    opcheck(NS.depthwise_conv3x3.default, (x, wd, 1, 1, False), test_utils=utils)
    opcheck(NS.interpolate_bilinear.default, (x, 17, 21, True), test_utils=utils)
    lo = to_dev_nhwc(rnd((1, 19, 5, 7), 4), torch.float32, pitch=24, off=0).requires_grad_()
    t = torch.randint(0, 19, (1, 17, 25), device=DEV)
    opcheck(NS.upsample_cross_entropy.default, (lo, t, 17, 25, -1, True), test_utils=utils)


def test_criss_cross_attention_op_matches_functional_and_gradcheck_shapes():
    """torch.ops.segmentron_hip.criss_cross_attention == functional.criss_cross_attention
    (same kernels), forward and every gradient."""
    from segmentron_amd import functional as F
    g = torch.Generator().manual_seed(2)
    mk = lambda c: torch.randn(2, 11, 14, c, generator=g).cuda()
    base = [mk(8), mk(8), mk(64), mk(64), torch.tensor([0.6]).cuda()]
    dout = mk(64)
    res = []
    for fn in (lambda *a: NS.criss_cross_attention(*a)[0], F.criss_cross_attention):
        ins = [t.clone().requires_grad_() for t in base]
        out = fn(*ins)
        out.backward(dout)
        res.append([out.detach()] + [t.grad for t in ins])
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_segmentation_counts_op():
    g = torch.Generator().manual_seed(3)
    logits = (torch.randn(2, 7, 33, 41, generator=g) * 3).cuda()
    tgt = torch.randint(-1, 7, (2, 33, 41), generator=g).cuda()
    cnt = NS.segmentation_counts(logits, tgt, 7).cpu()
    pred = logits.argmax(1).cpu()
    t = tgt.cpu()
    assert int(cnt[1]) == int((t >= 0).sum())
    assert int(cnt[2:9].sum()) == int(((pred == t) & (t >= 0)).sum())
