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

# ---- r02 additions: the stride-2 entry-flow layer and the ASPP wide-dilation layers
if len(sys.argv) > 2 and sys.argv[2] == "more":
    N2, H2, W2, C2 = 2, 513, 1025, 128            # block1 sep_conv3 input (stride 2)
    x2 = (torch.randn(N2, H2, W2, C2, device="cuda") * 1.3 + 0.4).to(dt)
    dy2 = torch.randn(N2, (H2 + 1) // 2, (W2 + 1) // 2, C2, device="cuda").to(dt)
    w2 = torch.randn(C2, 1, 3, 3, device="cuda") * 0.4
    s2_, t2_ = torch.rand(C2, device="cuda") + 0.5, torch.randn(C2, device="cuda") * 0.3
    w9 = w2.view(C2, 9).t().contiguous()
    timed("dw s2 fwd (strip) 128ch@513x1025", lambda: K.dwconv(x2, w9, 2, 1, (3, s2_, t2_), want_stats=True))
    timed("dw s2 fused bwd 128ch@513x1025", lambda: K.dwconv_bwd_fused_s2(x2, dy2, w2, (3, s2_, t2_), want_bn=True))
    N3, H3, W3, C3 = 2, 65, 129, 2048             # ASPP branches
    x3 = (torch.randn(N3, H3, W3, C3, device="cuda") * 1.3 + 0.4).to(dt)
    dy3 = torch.randn(N3, H3, W3, C3, device="cuda").to(dt)
    w3 = (torch.randn(9, C3, device="cuda") * 0.4).contiguous()
    for dil in (6, 12, 18):
        timed("dw dil %d fwd (row chain) 2048ch@65x129" % dil,
              lambda: K.dwconv(x3, w3, 1, dil, (1, None, None), want_stats=True))
        timed("dw dil %d fused bwd (row chain)" % dil,
              lambda: K.dwconv_bwd_fused(x3, dy3, w3, dil, (1, None, None), want_bn=False))
