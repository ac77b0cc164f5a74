"""Runs the memory-bound kernels of the middle flow (depthwise fwd / fused bwd, BatchNorm
backward apply / reduce, materialise) at the C3 shape [2,65,129,728] bf16 a few times — the
target of rocprofv3 --pmc passes (tools/lab/pmc_lab.sh) and of quick timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segmentron_amd import hip_ops as K

it = int(sys.argv[1]) if len(sys.argv) > 1 else 5
N, H, W, C = 2, 65, 129, 728
dt = torch.bfloat16
x = (torch.randn(N, H, W, C, device="cuda") * 1.3 + 0.4).to(dt)
dy = torch.randn(N, H, W, C, device="cuda").to(dt)
w = torch.randn(C, 1, 3, 3, device="cuda") * 0.4
s, t = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.3
c0, c1 = torch.randn(C, device="cuda") * 0.1, torch.randn(C, device="cuda") * 0.1
ev = [torch.cuda.Event(enable_timing=True) for _ in range(12)]
def timed(name, fn):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    print("%-28s %7.1f us" % (name, a.elapsed_time(b) * 1e3 / it))
timed("dw fwd (+stats)", lambda: K.dwconv(x, w, 1, 1, (3, s, t), want_stats=True))
timed("dw bwd fused (+bn sums)", lambda: K.dwconv_bwd_fused(x, dy, w, 1, (3, s, t), want_bn=True, torch_layout=True))
timed("bn_bwd_apply (affine)", lambda: K.bn_bwd_apply(dy, x, (2, s, t), c0, c1))
timed("bn_bwd_reduce_partial", lambda: K.bn_bwd_reduce_partial(dy, x, (3, s, t)))
timed("bn_apply (+residual)", lambda: K.bn_apply(x, (2, s, t), dy, None))
