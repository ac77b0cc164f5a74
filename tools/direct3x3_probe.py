"""Runs the direct halo-tile 3x3 kernels (csrc/conv3x3_direct.hip) at the xception conv2 shape
[2,513,1025,32] -> 64 bf16 — forward (BN+ReLU prologue, statistics), data gradient (64 -> 32),
weight gradient — a few times: the target of rocprofv3 --pmc passes (tools/lab/pmc_lab.sh) and
of quick timing.  SEG_NO_DIRECT=1 runs the same calls on the implicit-GEMM kernels instead."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segmentron_amd import functional as F, hip_ops as K
from segmentron_amd._lib import LIB

it = int(sys.argv[1]) if len(sys.argv) > 1 else 5
if os.environ.get("SEG_NO_DIRECT") == "1":
    LIB.query("seg_conv_gemm_px256", 2 | 4)
    LIB.query("seg_conv_gemm_wgrad_config", 4)
N, H, W, C, O = 2, 513, 1025, 32, 64
dt = torch.bfloat16
x = (torch.randn(N, H, W, C, device="cuda") * 1.3 + 0.4).to(dt)
dy = torch.randn(N, H, W, O, device="cuda").to(dt)
w = torch.randn(O, C, 3, 3, device="cuda") * 0.1
s, t = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.3
wp = F.pack_conv_weight(w, C, dt)
wt = F.pack_conv_weight_dgrad(w, O, dt)


def timed(name, fn, flop):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / it
    print("%-34s %7.1f us  %6.0f TFLOP/s" % (name, us, flop / us / 1e6))


fl = 2.0 * N * H * W * 9 * C * O
timed("conv2 fwd (+BN/ReLU prologue, stats)", lambda: K.conv_gemm(x, wp, O, 3, 3, 1, 1, 1, (3, s, t), None, None, True), fl)
timed("conv2 dgrad (64 -> 32)", lambda: K.conv_gemm(dy, wt, C, 3, 3, 1, 1, 1), fl)
timed("conv2 wgrad (+prologue)", lambda: K.conv_wgrad(x, dy, O, 3, 3, 1, 1, 1, (3, s, t)), fl)
