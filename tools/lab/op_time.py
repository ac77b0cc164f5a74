#!/usr/bin/env python3
"""Per-op timing of the C3 middle-flow shapes through the C-ABI (kernel A/B experiments):

    SEGMENTRON_HIP_LIB=tools/lab/variants/libX.so python tools/lab/op_time.py [ops...]

Every op is run `--iters` times back to back between two events (the inputs of one call are the
24 MB tensors a step just produced, i.e. Infinity-Cache resident, as in the train step)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmentron_amd import hip_ops as K  # noqa: E402
from segmentron_amd.graph import capture  # noqa: E402

DEV = "cuda"


def timeit(fn, iters):
    """`iters` launches captured into ONE HIP graph (the host cannot keep up with 5-30 us kernels
    launched one by one); the replay is timed."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
        torch.cuda.synchronize()
        with capture(graph, stream=side):
            for _ in range(iters):
                fn()
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("ops", nargs="*")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--shape", default="2,65,129,728")
    args = ap.parse_args()
    N, H, W, C = (int(v) for v in args.shape.split(","))
    dt = torch.bfloat16
    g = torch.Generator(device=DEV).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=DEV, generator=g)
    x = rn(N, H, W, C).to(dt)
    dy = rn(N, H, W, C).to(dt)
    res = rn(N, H, W, C).to(dt)
    wdw = rn(C, 1, 3, 3) * 0.3
    sc, sh = torch.rand(C, device=DEV) + 0.5, rn(C) * 0.1
    pro = (3, sc, sh)
    wpw = (rn(C, C) * 0.04).to(dt).contiguous()
    c0, c1 = rn(C) * 0.01, rn(C) * 0.01
    M = N * H * W
    ops = {}
    ops["dw_fwd"] = lambda: K.dwconv(x, wdw, 1, 1, pro, None, True)
    w9c = wdw.reshape(C, 9).t().contiguous()  # tap-major packing (functional.pack_dw_weight)
    ops["dw_fwd_s2"] = lambda: K.dwconv(x, w9c, 2, 1, pro, None, True)
    ops["dw_bwd"] = lambda: K.dwconv_bwd_fused(x, dy, wdw, 1, pro, want_bn=True, torch_layout=True,
                                               raw_dw=True)
    ops["dw_bwd_res"] = lambda: K.dwconv_bwd_fused(x, dy, wdw, 1, (1, None, None), want_bn=False,
                                                   torch_layout=True, raw_dw=True, res=res)
    ops["bn_bwd_apply"] = lambda: K.bn_bwd_apply(dy, x, (2, sc, sh), c0, c1, out=dy)
    ops["bn_apply_res"] = lambda: K.bn_apply(x, (2, sc, sh), res, (0, None, None))
    ops["bn_bwd_reduce"] = lambda: K.bn_bwd_reduce_partial(dy, x, (2, sc, sh), None, None)
    ops["gemm_fwd"] = lambda: K.conv_gemm(x, wpw, C, 1, 1, 1, 0, 1, None, None, None, True)
    ops["gemm_dgrad"] = lambda: K.conv_gemm(dy, wpw, C, 1, 1, 1, 0, 1, ep=(x, c0, c1))
    ops["wgrad"] = lambda: K.conv_wgrad(x, dy, C, 1, 1, 1, 0, 1, None, raw_partial=True)
    wf = rn(C, C) * 0.04
    dwp = K.conv_wgrad(x, dy, C, 1, 1, 1, 0, 1, None, raw_partial=True)
    ops["fold_bwd_reduce"] = lambda: K.fold_bwd_reduce(wf, dwp, sc, sh, None)
    ops["fold_weights"] = lambda: K.fold_weights(wf, sc, sh, dt, want_transpose=True)
    _, part = K.conv_gemm(x, wpw, C, 1, 1, 1, 0, 1, None, None, None, True)
    bnw, bnb = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    ops["bn_finalize_p"] = lambda: K.bn_finalize_p(part, float(M), bnw, bnb, 1e-3, 0.1, rm, rv)
    _, pw_, pb_ = K.dwconv_bwd_fused(x, dy, wdw, 1, pro, want_bn=True, torch_layout=True, raw_dw=True)
    mean, invstd = rn(C) * 0.1, torch.rand(C, device=DEV) + 0.5
    ops["dw_bwd_finalize"] = lambda: K.dw_bwd_finalize(pb_, pw_, float(M), mean, invstd, bnw)
    ops["bn_bwd_finalize_p"] = lambda: K.bn_bwd_finalize_p(pb_, float(M), mean, invstd, bnw)
    def three():
        g_, pw2, pb2 = K.dwconv_bwd_fused(x, dy, wdw, 1, pro, want_bn=True, torch_layout=True, raw_dw=True)
        _, _, c0_, c1_, _ = K.dw_bwd_finalize(pb2, pw2, float(M), mean, invstd, bnw)
        K.bn_bwd_apply(g_, x, (2, sc, sh), c0_, c1_, out=g_)
    ops["dw_bwd+fin+apply"] = three

    def red_fin():
        pr = K.bn_bwd_reduce_partial(dy, x, (2, sc, sh), None, None)
        K.bn_bwd_finalize_p(pr, float(M), mean, invstd, bnw)
    ops["bn_bwd_reduce+fin"] = red_fin
    # what does one extra tiny launch cost inside a graph?  (pair - single)
    def pair():
        K.conv_gemm(x, wpw, C, 1, 1, 1, 0, 1, None, None, None, True)
        K.bn_finalize_p(part, float(M), bnw, bnb, 1e-3, 0.1, rm, rv)
    ops["gemm_fwd+fin"] = pair

    def triple():
        K.conv_gemm(x, wpw, C, 1, 1, 1, 0, 1, None, None, None, True)
        K.bn_finalize_p(part, float(M), bnw, bnb, 1e-3, 0.1, rm, rv)
        K.fold_weights(wf, sc, sh, dt, want_transpose=True)
    ops["gemm_fwd+fin+fold"] = triple
    which = args.ops or list(ops)
    for name in which:
        us = timeit(ops[name], args.iters)
        print("%-18s %8.2f us" % (name, us), flush=True)


if __name__ == "__main__":
    main()
