#!/usr/bin/env python3
"""Micro-benchmark of the depthwise / BatchNorm kernels on DeepLabv3+/xception65 shapes
(GPU box).  Prints microseconds and effective algorithmic GB/s (2 B * elements touched)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmentron_amd import hip_ops as K  # noqa: E402

SHAPES = [("middle 728 @65x129", 2, 65, 129, 728), ("entry 128 @513x1025", 2, 513, 1025, 128),
          ("exit 1536 @65x129 d2", 2, 65, 129, 1536), ("decoder 304 @257x513", 2, 257, 513, 304)]


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", type=int, default=-1)
    ap.add_argument("--what", default="all")
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    dt = torch.bfloat16
    for idx, (name, N, H, W, C) in enumerate(SHAPES):
        if args.only >= 0 and idx != args.only:
            continue
        dil = 2 if "d2" in name else 1
        x = torch.randn((N, H, W, C), device="cuda").to(dt)
        dy = torch.randn((N, H, W, C), device="cuda").to(dt)
        w9c = torch.randn((9, C), device="cuda") * 0.3
        s = torch.rand(C, device="cuda") + 0.5
        t = torch.randn(C, device="cuda") * 0.1
        pro = (3, s, t)
        mb = N * H * W * C * 2 / 1e6
        r = {"shape": name, "tensor_MB": round(mb, 1)}

        def rec(key, us, passes):
            r[key] = {"us": round(us, 1), "GBps": round(passes * mb / us * 1e3, 0)}
        rec("fwd+stats", timeit(lambda: K.dwconv(x, w9c, 1, dil, pro, want_stats=True), args.iters), 2)
        if args.what == "fwd":
            print(json.dumps(r), flush=True)
            continue
        rec("dgrad", timeit(lambda: K.dwconv_dgrad(dy, w9c, 1, dil, (H, W))), 2)
        rec("wgrad", timeit(lambda: K.dwconv_wgrad(x, dy, 1, dil, pro)), 2)
        rec("bn_reduce", timeit(lambda: K.bn_bwd_reduce_partial(dy, x, pro)), 2)
        rec("bwd_fused", timeit(lambda: K.dwconv_bwd_fused(x, dy, w9c, dil, pro, want_bn=True)), 3)
        rec("bn_bwd_apply", timeit(lambda: K.bn_bwd_apply(dy, x, pro, s, t)), 3)
        rec("bn_apply_res", timeit(lambda: K.bn_apply(x, pro, dy, None)), 3)
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
