#!/usr/bin/env python3
"""Debug (GPU box): PSP head in isolation — HIP modules vs torch CPU float64 on the same c4."""
import os
import sys

import torch
import torch.nn.functional as TF

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import segmentron_amd  # noqa: E402
from segmentron_amd import functional as F  # noqa: E402
from segmentron_amd.config import cfg, reset_cfg  # noqa: E402
from segmentron_amd.models.pspnet import _PSPHead  # noqa: E402
from oracle import synth, torch_ref  # noqa: E402

reset_cfg()
segmentron_amd.set_compute_dtype(torch.float32)
torch.manual_seed(0)
head = _PSPHead(19)
sd = synth.synth_like(head.state_dict(), seed=3)
head.load_state_dict(sd)
head = head.cuda().train()
head.block[3].p = 0.0
N, H, W = 2, 7, 9
c4 = torch.relu(torch.randn(N, 2048, H, W, dtype=torch.float64))
g = torch.randn(N, 19, H, W, dtype=torch.float64)

from segmentron_amd import hip_ops as K  # noqa: E402
calls = []
_orig = K.bn_bwd_reduce_partial


def _rec(g, x, pro, chan_mul=None, elem_mul=None):
    calls.append((g.clone(), x.clone(), pro))
    return _orig(g, x, pro, chan_mul, elem_mul)


K.bn_bwd_reduce_partial = _rec
x_dev = c4.float().permute(0, 2, 3, 1).contiguous().cuda().requires_grad_()
y = head(F.Act(x_dev))  # NHWC [N,H,W,19]
y.backward(g.float().permute(0, 2, 3, 1).contiguous().cuda())

# reference
osd = {("head." + k): (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
osd = torch_ref.clone_state(osd, requires_grad=True)
net = torch_ref.OracleNet(osd, training=True, drop_p=0.0)
xr = c4.clone().requires_grad_()
feats = [xr]
for i, o in enumerate((1, 2, 3, 6)):
    f = TF.adaptive_avg_pool2d(xr, o)
    f = net.conv_bn_relu(f, "head.psp.convs.%d" % i)
    feats.append(TF.interpolate(f, (H, W), mode="bilinear", align_corners=True))
cat = torch.cat(feats, 1)
cat.retain_grad()
z = net.conv(cat, "head.block.0", 1, 1)
z.retain_grad()
bno = net.bn(z, "head.block.1")
act = torch.relu(bno)
act.retain_grad()
yr = net.conv(act, "head.block.4")
yr.backward(g)


def rel(a, b):
    return ((a.double().cpu() - b).norm() / b.norm()).item()


print("fwd", rel(y.detach().permute(0, 3, 1, 2), yr.detach()))
print("d c4", rel(x_dev.grad.permute(0, 3, 1, 2), xr.grad))
for k, p in head.named_parameters():
    print("%-28s %.3e" % (k, rel(p.grad, osd["head." + k].grad)))

g0, x0, pro0 = [c for c in calls if c[0].shape[-1] == 512][0]
print("first BN-backward call: g shape", tuple(g0.shape), "mode", pro0[0])
print("  g vs ref act.grad", rel(g0.permute(0, 3, 1, 2), act.grad), " x vs ref z", rel(x0.permute(0, 3, 1, 2), z.detach()))
zm = z.detach().mean((0, 2, 3)); zv = z.detach().var((0, 2, 3), unbiased=False)
sc_ref = osd["head.head.block.1.weight"].detach() / torch.sqrt(zv + 1e-5) if "head.head.block.1.weight" in osd else osd["head.block.1.weight"].detach() / torch.sqrt(zv + 1e-5)
print("  scale vs ref", rel(pro0[1], sc_ref))
colsum_g = g0.double().cpu().sum((0, 1, 2))
print("  colsum(g) vs ref", rel(colsum_g, act.grad.sum((0, 2, 3))))
