"""criss-cross attention (csrc/cca.hip) — the reference's CUDA extension
segmentron/modules/csrc/criss_cross_attention/ca_cuda.cu re-done for gfx950.
CPU: the oracle's tensor-algebra restatement against a statement-by-statement transcription of
the kernel source.  GPU: the fused HIP op (forward + all five gradients) against the oracle in
fp64 through autograd."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import torch_ref as R  # noqa: E402


@pytest.mark.parametrize("shape", [(2, 3, 4, 5), (1, 2, 5, 3), (1, 1, 1, 4), (1, 2, 3, 1)])
def test_oracle_algebra_equals_kernel_source_loops(shape):
    N, C, H, W = shape
    g = torch.Generator().manual_seed(0)
    t = torch.randn(N, C, H, W, generator=g, dtype=torch.float64)
    f = torch.randn(N, C, H, W, generator=g, dtype=torch.float64)
    w = R.cca_weight(t, f)
    assert tuple(w.shape) == (N, H + W - 1, H, W)
    assert (w - R.cca_weight_loops(t, f)).abs().max() < 1e-12
    a = torch.softmax(w, 1)
    v = torch.randn(N, 4, H, W, generator=g, dtype=torch.float64)
    assert (R.cca_map(a, v) - R.cca_map_loops(a, v)).abs().max() < 1e-12


def _nhwc(t, dtype):
    return t.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 9, 13, 64), (1, 49, 49, 64), (1, 5, 70, 128), (2, 33, 17, 512)])
def test_criss_cross_attention_fwd_bwd_matches_oracle(shape, dtype):
    """out = gamma * CCA(q, k, v) + x and d/d{q, k, v, x, gamma} (H + W - 1 up to 97: several
    attended entries per lane; C' = C / 8 as in the module)."""
    from segmentron_amd import functional as F
    N, H, W, C = shape
    Cq = max(C // 8, 8)
    g = torch.Generator().manual_seed(1)
    mk = lambda c, s: (torch.randn(N, c, H, W, generator=g) * s)
    q, k, v, x = mk(Cq, 0.7), mk(Cq, 0.7), mk(C, 1.0), mk(C, 1.0)
    gamma = torch.tensor([0.7])
    dout = mk(C, 1.0)
    quant = (lambda t: t.to(dtype).float()) if dtype == torch.bfloat16 else (lambda t: t)
    q, k, v, x, dout = [quant(t) for t in (q, k, v, x, dout)]
    # oracle, fp64
    ref_in = [t.double().requires_grad_() for t in (q, k, v, x, gamma)]
    att = torch.softmax(R.cca_weight(ref_in[0], ref_in[1]), 1)
    ref = ref_in[4] * R.cca_map(att, ref_in[2]) + ref_in[3]
    ref.backward(dout.double())
    # HIP
    dq, dk, dv, dx = [_nhwc(t, dtype).requires_grad_() for t in (q, k, v, x)]
    dg = gamma.clone().cuda().requires_grad_()
    out = F.criss_cross_attention(dq, dk, dv, dx, dg)
    out.backward(_nhwc(dout, dtype))
    tol = 2e-5 if dtype == torch.float32 else 1.2e-2

    def close(got_nhwc, want_nchw, what):
        got = got_nhwc.detach().float().cpu().permute(0, 3, 1, 2).double()
        err = (got - want_nchw).abs().max().item()
        scale = want_nchw.abs().max().item()
        assert err <= tol * scale, "%s: max err %.3e vs scale %.3e" % (what, err, scale)
    close(out, ref.detach(), "out")
    close(dq.grad, ref_in[0].grad, "dq")
    close(dk.grad, ref_in[1].grad, "dk")
    close(dv.grad, ref_in[2].grad, "dv")
    close(dx.grad, ref_in[3].grad, "dx")
    # d gamma = <dout, CCA> is a sum of N*H*W*C products of both signs; the bf16 path stores CCA
    # rounded to bf16 (2^-9 relative, independent per element), so its floor is a random walk
    # over the terms — bound: 4 sigma of that walk; fp32: 1e-4 of the same scale
    terms = (dout.double() * ((ref.detach() - ref_in[3].detach()) / 0.7)).flatten()
    sigma = terms.pow(2).sum().sqrt().item()
    bound = 1e-4 * sigma if dtype == torch.float32 else 4 * 2.0 ** -9 * sigma
    gg, gr = dg.grad.item(), ref_in[4].grad.item()
    assert abs(gg - gr) <= bound + 1e-6, (gg, gr, bound)


@pytest.mark.gpu
def test_attention_rows_are_probabilities_and_map_transposes():
    """softmax rows sum to 1; <map(w, b), c> == <b, map^T(w, c)> (the transposed gather really
    is the adjoint of the forward gather — an identity independent of any oracle)."""
    from segmentron_amd import hip_ops as K
    g = torch.Generator().manual_seed(5)
    N, H, W, C = 2, 21, 30, 32
    q = torch.randn(N, H, W, 8, generator=g).cuda()
    k = torch.randn(N, H, W, 8, generator=g).cuda()
    att = K.cca_attention(q, k)
    assert tuple(att.shape) == (N, H, W, H + W - 1)
    assert (att.sum(-1) - 1).abs().max().item() < 1e-5 and att.min().item() >= 0
    b = torch.randn(N, H, W, C, generator=g).cuda()
    c = torch.randn(N, H, W, C, generator=g).cuda()
    lhs = (K.cca_map(att, b).double() * c.double()).sum()
    rhs = (b.double() * K.cca_map(att, c, transposed=True).double()).sum()
    assert abs(lhs.item() - rhs.item()) <= 1e-5 * abs(lhs.item())


# ---------------------------------------------------------------- pinned to the reference's code
def _ref_vectors():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "cca_ref_vectors.npz"))


def _check_against(ref_fns, shape, dtype, tol):
    """oracle forward + torch autograd of the oracle == the reference kernels' forward and their
    four hand-written backward kernels, on the inputs of oracle/gen_golden_cca.py."""
    from oracle.gen_golden_cca import inputs
    t, f, v, att, dwt, dout = inputs(shape, dtype)
    weight, (dt_, df_), agg, (dw_, dg_) = ref_fns(t, f, v, att, dwt, dout)
    scale = lambda x: max(x.abs().max().item(), 1e-30)
    tq, fq = t.clone().requires_grad_(), f.clone().requires_grad_()
    w = R.cca_weight(tq, fq)
    assert (w.detach() - weight).abs().max().item() <= tol * scale(weight)
    w.backward(dwt)      # ca_backward_kernel_t / _f  (ca_cuda.cu:38-94)
    assert (tq.grad - dt_).abs().max().item() <= tol * scale(dt_)
    assert (fq.grad - df_).abs().max().item() <= tol * scale(df_)
    aq, vq = att.clone().requires_grad_(), v.clone().requires_grad_()
    o = R.cca_map(aq, vq)
    assert (o.detach() - agg).abs().max().item() <= tol * scale(agg)
    o.backward(dout)     # ca_map_backward_kernel_w / _g  (ca_cuda.cu:121-177)
    assert (aq.grad - dw_).abs().max().item() <= tol * scale(dw_)
    assert (vq.grad - dg_).abs().max().item() <= tol * scale(dg_)


@pytest.mark.parametrize("dtype,tag,tol", [(torch.float64, "f64", 1e-13), (torch.float32, "f32", 2e-6)])
def test_oracle_matches_vectors_produced_by_the_compiled_reference_kernels(dtype, tag, tol):
    """the oracle's criss-cross attention is pinned to numbers that the
    reference's OWN kernel source produced (ca_cuda.cu compiled as host C++ by
    oracle/cca_ref/build.sh; vectors by oracle/gen_golden_cca.py) — forward energies, the
    aggregation, and the reference's four backward kernels vs torch autograd of the oracle."""
    g = _ref_vectors()
    for i, shape in enumerate(g["shapes"].tolist()):
        get = lambda k: torch.from_numpy(g["%d_%s_%s" % (i, tag, k)])
        _check_against(lambda *a: (get("weight"), (get("dt"), get("df")), get("agg"),
                                   (get("dw"), get("dg"))), tuple(shape), dtype, tol)


def test_oracle_matches_the_compiled_reference_kernels_live():
    """Same comparison against oracle/_ref/libcca_ref.so itself on further shapes (skipped where
    neither the built library nor the reference checkout to build it from exists)."""
    from oracle import cca_ref
    if not cca_ref.available():
        pytest.skip("oracle/_ref/libcca_ref.so not built and no reference checkout here")
    for shape in [(2, 5, 7, 9), (1, 3, 34, 5), (1, 2, 6, 37), (2, 4, 1, 1)]:
        _check_against(lambda t, f, v, att, dwt, dout: (
            cca_ref.ca_forward(t, f), cca_ref.ca_backward(dwt, t, f),
            cca_ref.ca_map_forward(att, v), cca_ref.ca_map_backward(dout, att, v)),
            shape, torch.float64, 1e-13)
