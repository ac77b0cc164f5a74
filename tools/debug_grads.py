#!/usr/bin/env python3
"""Debug helper (GPU box): per-parameter gradient error of the HIP fp32 path vs the fp64 oracle."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_more_models as T  # noqa: E402
from oracle import synth  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "c4"
c = T.CASES[tag]
model, sd = T._build_hip(tag, torch.float32, True)
H, W = c["hw"]
x = synth.synth_images(2, H, W, seed=0)
y = synth.synth_targets(2, H, W, seed=0)
outs = model(x.cuda())
loss = torch.nn.functional.cross_entropy(outs[0], y.cuda(), ignore_index=-1)
for o in outs[1:]:
    loss = loss + c["aux_weight"] * torch.nn.functional.cross_entropy(o, y.cuda(), ignore_index=-1)
loss.backward()
_, _, g64 = T._oracle(tag, sd, x, True, torch.float64, y)
_, _, g32 = T._oracle(tag, sd, x, True, torch.float32, y)
params = dict(model.named_parameters())
rows = []
for k, t64 in g64.items():
    gh = params[k].grad.detach().cpu().double()
    n = t64.norm().item()
    rows.append((k, (gh - t64).norm().item() / max(n, 1e-30), (g32[k].double() - t64).norm().item() / max(n, 1e-30), n))
sel = [r for r in rows if not r[0].startswith("encoder.layer") or ".0." in r[0] and "conv1" in r[0]]
for r in sel:
    print("%-48s hip %.2e cpu32 %.2e |g| %.2e" % r)
