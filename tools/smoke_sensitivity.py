"""The smoke fixture's own sensitivity (CPU only): how far the float64 oracle's gradient of the
DeepLabv3+/xception65 step at 65x97 (seed 1, conditioned state) moves under 1e-6 .. 1e-5 relative
input noise, next to the float32 oracle under the same noise.  One ReLU input of the exit flow sits
within ~1e-6 of zero: perturbations that cross it move the gradient by 1.3e-3 global-rel
(__graft_entry__.smoke)."""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import synth, torch_ref
import segmentron_amd
from segmentron_amd.config import cfg, reset_cfg
from conftest import C3_OVERRIDES
reset_cfg(); cfg.update_from_list(C3_OVERRIDES); cfg.PHASE = "test"; cfg.check_and_freeze()
model = segmentron_amd.get_segmentation_model()
sd = synth.synth_like(model.state_dict(), seed=1, conditioned=True)
H, W = 65, 97
x = synth.synth_images(2, H, W, seed=1); y = synth.synth_targets(2, H, W, seed=1)
def grads(dt, xin):
    s = {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    osd = torch_ref.clone_state(s, requires_grad=True)
    net = torch_ref.OracleNet(osd, training=True, drop_p=0.0)
    ref = net.deeplabv3_plus_xception65(xin.to(dt)); rl = torch_ref.mix_softmax_ce(ref, y); rl.backward()
    return {k: v.grad.double() for k, v in osd.items() if v.grad is not None}
g64 = grads(torch.float64, x)
def rel(g):
    e = sum((g[k] - t).norm().item() ** 2 for k, t in g64.items()); d = sum(t.norm().item() ** 2 for t in g64.values())
    return (e / d) ** 0.5
for amp in (0.0, 1e-6, 3e-6, 1e-5):
    for seed in range(3 if amp else 1):
        gen = torch.Generator().manual_seed(100 + seed)
        xp = x * (1 + amp * torch.randn(x.shape, generator=gen))
        # perturbed fp64 (true sensitivity to the perturbation incl. mask flips) and perturbed fp32
        print("amp %.0e seed %d: fp64(perturbed) vs fp64 %.3e | fp32(perturbed) vs fp64 %.3e"
              % (amp, seed, rel(grads(torch.float64, xp)), rel(grads(torch.float32, xp))), flush=True)
