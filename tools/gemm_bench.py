#!/usr/bin/env python3
"""Micro-benchmark of the MFMA implicit-GEMM kernels on the dominant DeepLabv3+/xception65 shapes
(run on the GPU box, optionally under `rocprofv3 --pmc ...`).  Prints TFLOP/s per shape."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmentron_amd import hip_ops as K  # noqa: E402

SHAPES = [  # (name, N, H, W, C, O)
    ("middle 728->728 @65x129", 2, 65, 129, 728, 728),
    ("exit 1536->2048 @65x129", 2, 65, 129, 1536, 2048),
    ("decoder 304->256 @257x513", 2, 257, 513, 304, 256),
    ("entry 128->128 @513x1025", 2, 513, 1025, 128, 128),
]


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--only", type=int, default=-1)
    ap.add_argument("--wgrad", action="store_true")
    args = ap.parse_args()
    from segmentron_amd._lib import LIB
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    out = []
    for i, (name, N, H, W, C, O) in enumerate(SHAPES):
        if args.only >= 0 and i != args.only:
            continue
        x = torch.randn((N, H, W, C), device="cuda").to(dt)
        w = (torch.randn((O, C), device="cuda") * 0.05).to(dt)
        dy = torch.randn((N, H, W, O), device="cuda").to(dt)
        s = torch.rand(C, device="cuda") + 0.5
        t = torch.randn(C, device="cuda") * 0.1
        flop = 2.0 * N * H * W * C * O
        r = {"shape": name, "GFLOP": flop / 1e9}
        r["fwd_plain_TF"] = flop / timeit(lambda: K.conv_gemm(x, w, O, 1, 1, 1, 0, 1), args.iters) / 1e12
        r["fwd_stats_TF"] = flop / timeit(
            lambda: K.conv_gemm(x, w, O, 1, 1, 1, 0, 1, want_stats=True), args.iters) / 1e12
        r["fwd_prologue_stats_TF"] = flop / timeit(
            lambda: K.conv_gemm(x, w, O, 1, 1, 1, 0, 1, (3, s, t), want_stats=True), args.iters) / 1e12
        if args.wgrad:
            r["wgrad_plain_TF"] = flop / timeit(
                lambda: K.conv_wgrad(x, dy, O, 1, 1, 1, 0, 1), args.iters) / 1e12
            r["wgrad_prologue_TF"] = flop / timeit(
                lambda: K.conv_wgrad(x, dy, O, 1, 1, 1, 0, 1, (3, s, t)), args.iters) / 1e12
        print(json.dumps(r), flush=True)
        out.append(r)


if __name__ == "__main__":
    main()
