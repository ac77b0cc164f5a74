"""GPU bisect: BatchNorm-backward sums of the fused depthwise backward (bf16 vs fp32) against a
float64 torch reference on the same (bf16-representable) inputs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as TF
from segmentron_amd import hip_ops as K

torch.manual_seed(0)
N, H, W, C = 2, 65, 129, 728
for dtype in (torch.float32, torch.bfloat16):
    for dil in (1, 2):
        x = (torch.randn(N, H, W, C, device="cuda") * 1.3 + 0.4).to(dtype)
        dy = torch.randn(N, H, W, C, device="cuda").to(dtype)
        w = (torch.randn(C, 1, 3, 3, device="cuda") * 0.4)
        s = (torch.rand(C, device="cuda") + 0.5)
        t = torch.randn(C, device="cuda") * 0.3
        g, dW, pb = K.dwconv_bwd_fused(x, dy, w, dil, (3, s, t), want_bn=True, torch_layout=True)
        sums = K.colsum(pb).double()
        # reference
        xd = x.double().permute(0, 3, 1, 2).requires_grad_()
        a = torch.relu(xd * s.double().view(1, -1, 1, 1) + t.double().view(1, -1, 1, 1))
        a.retain_grad()
        y = TF.conv2d(a, w.double(), None, 1, dil, dil, groups=C)
        y.backward(dy.double().permute(0, 3, 1, 2))
        gref = (a.grad * (a > 0)).detach()            # masked dgrad wrt act(x)
        xr = x.double().permute(0, 3, 1, 2)
        r1, r2 = gref.sum((0, 2, 3)), (gref * xr).sum((0, 2, 3))
        a1, a2 = gref.abs().sum((0, 2, 3)), (gref * xr).abs().sum((0, 2, 3))
        e1 = ((sums[:C] - r1).abs() / a1).max().item()
        e2 = ((sums[C:] - r2).abs() / a2).max().item()
        rel1 = ((sums[:C] - r1).norm() / r1.norm()).item()
        rel2 = ((sums[C:] - r2).norm() / r2.norm()).item()
        eg = ((g.double().permute(0, 3, 1, 2) - gref).norm() / gref.norm()).item()
        # dgamma-like combination with mean/invstd of x
        mu, var = xr.mean((0, 2, 3)), xr.var((0, 2, 3), unbiased=False)
        dg_ref = (r2 - mu * r1) / var.sqrt()
        dg = (sums[C:] - mu * sums[:C]) / var.sqrt()
        print("%s dil %d: g L2-rel %.2e | sum g' err/abs-sum %.2e (L2-rel %.2e) | sum g'x err/abs-sum %.2e "
              "(L2-rel %.2e) | dgamma-like L2-rel %.2e" % (str(dtype)[6:], dil, eg, e1, rel1, e2, rel2,
              ((dg - dg_ref).norm() / dg_ref.norm()).item()))
