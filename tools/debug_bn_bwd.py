#!/usr/bin/env python3
"""Debug (GPU box): conv3x3(4096->512) -> BN(train) -> ReLU -> 1x1(512->19, bias); every
intermediate of the HIP backward vs torch CPU float64."""
import os
import sys

import torch
import torch.nn.functional as TF

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from segmentron_amd import functional as F, hip_ops as K  # noqa: E402

torch.manual_seed(0)
N, H, W, C, O = 2, 7, 9, 4096, 512
x = torch.relu(torch.randn(N, C, H, W, dtype=torch.float64))
w0 = torch.randn(O, C, 3, 3, dtype=torch.float64) * (2.0 / (C * 9)) ** 0.5
gamma, beta = torch.rand(O, dtype=torch.float64) + 0.5, torch.randn(O, dtype=torch.float64) * 0.1
w1 = torch.randn(19, O, 1, 1, dtype=torch.float64) * 0.05
b1 = torch.randn(19, dtype=torch.float64) * 0.1
g = torch.randn(N, 19, H, W, dtype=torch.float64)


def rel(a, b):
    a, b = a.double().cpu(), b.double()
    return ((a - b).norm() / b.norm()).item()


def nhwc(t):
    return t.float().permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.detach().cpu().permute(0, 3, 1, 2)


# reference
z = TF.conv2d(x, w0, None, 1, 1)
z.requires_grad_()
bnout = TF.batch_norm(z, None, None, gamma, beta, True, 0.1, 1e-5)
bnout.retain_grad()
a = torch.relu(bnout)
a.retain_grad()
y = TF.conv2d(a, w1, b1)
y.backward(g)
print("ref: |z| mean %.3f std %.3f ; per-channel |mean|/std median %.2f" % (
    z.mean().item(), z.std().item(), (z.mean((0, 2, 3)).abs() / z.std((0, 2, 3))).median().item()))

# HIP pieces
xd = nhwc(x)
wp = F.pack_conv_weight(w0.float().cuda(), C, torch.float32)
zd, partial = K.conv_gemm(xd, wp, O, 3, 3, 1, 1, 1, None, None, None, True)
print("z", rel(nchw(zd), z))
mean, invstd, scale, shift = K.bn_finalize_p(partial, N * H * W, gamma.float().cuda(), beta.float().cuda(), 1e-5, 0.1, None, None)
zm = z.detach().mean((0, 2, 3)); zv = z.detach().var((0, 2, 3), unbiased=False)
print("mean", rel(mean, zm), "invstd", rel(invstd, 1 / torch.sqrt(zv + 1e-5)))
gd = nhwc(a.grad)  # true grad wrt relu output
pro = (3, scale, shift)
part = K.bn_bwd_reduce_partial(gd, zd, pro)
sums = K.colsum(part).cpu()
gp = (a.grad * (bnout.detach() > 0))
print("s1", rel(sums[:O], gp.sum((0, 2, 3))), "s2", rel(sums[O:], (gp * z.detach()).sum((0, 2, 3))))
dgamma, dbeta, c0, c1 = K.bn_bwd_finalize_p(part, N * H * W, mean, invstd, gamma.float().cuda())
dgam_ref = (gp * (z.detach() - zm.view(1, -1, 1, 1)) / torch.sqrt(zv + 1e-5).view(1, -1, 1, 1)).sum((0, 2, 3))
print("dgamma", rel(dgamma, dgam_ref), "dbeta", rel(dbeta, gp.sum((0, 2, 3))))
dx = K.bn_bwd_apply(gd, zd, pro, c0, c1)
print("dx", rel(nchw(dx), z.grad))
# mask agreement
yl = zd * scale.view(1, 1, 1, -1) + shift.view(1, 1, 1, -1)
print("mask mismatches", int(((nchw(yl) > 0) != (bnout.detach() > 0)).sum()), "of", bnout.numel())
