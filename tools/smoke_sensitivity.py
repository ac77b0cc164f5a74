"""The smoke fixture's own sensitivity (CPU only): how far the float64 oracle's gradient of the
DeepLabv3+/xception65 step at __graft_entry__'s smoke fixture moves under 1e-6 .. 1e-5 relative
input noise.  A fixture is acceptable when the response is LINEAR in the noise (no ReLU input
within float32 rounding of zero whose mask flips: the r05 fixture, seed 1 at 65 x 97, jumped to
1.3e-3 under 1e-6) — asserted below, so that smoke() can hold the HIP float32 path to the plain
1e-3 bar (VERDICT r05 weak #1a).

    python tools/smoke_sensitivity.py [seed H W]      # default: the fixture smoke() uses
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import __graft_entry__ as G  # noqa: E402
import segmentron_amd  # noqa: E402
from conftest import C3_OVERRIDES  # noqa: E402
from oracle import synth, torch_ref  # noqa: E402
from segmentron_amd.config import cfg, reset_cfg  # noqa: E402


def main():
    seed, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (G.SMOKE_SEED, G.SMOKE_H, G.SMOKE_W)
    reset_cfg()
    cfg.update_from_list(C3_OVERRIDES)
    cfg.PHASE = "test"
    cfg.check_and_freeze()
    model = segmentron_amd.get_segmentation_model()
    sd = synth.synth_like(model.state_dict(), seed=seed, conditioned=True)
    x = synth.synth_images(2, H, W, seed=seed)
    y = synth.synth_targets(2, H, W, seed=seed)

    def grads(dt, xin):
        s = {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        osd = torch_ref.clone_state(s, requires_grad=True)
        net = torch_ref.OracleNet(osd, training=True, drop_p=0.0, momentum=1.0)
        torch_ref.mix_softmax_ce(net.deeplabv3_plus_xception65(xin.to(dt)), y).backward()
        return {k: v.grad.double() for k, v in osd.items() if v.grad is not None}

    g64 = grads(torch.float64, x)

    def rel(g):
        e = sum((g[k] - t).norm().item() ** 2 for k, t in g64.items())
        d = sum(t.norm().item() ** 2 for t in g64.values())
        return (e / d) ** 0.5

    print("fixture seed %d, %d x %d: float32 oracle vs float64 %.3e" % (seed, H, W, rel(grads(torch.float32, x))))
    worst = {}
    for amp in (1e-6, 1e-5):
        for s_ in range(3):
            gen = torch.Generator().manual_seed(100 + s_)
            xp = x * (1 + amp * torch.randn(x.shape, generator=gen))
            r = rel(grads(torch.float64, xp))
            worst[amp] = max(worst.get(amp, 0.0), r)
            print("  input noise %.0e, noise seed %d: float64(perturbed) vs float64 %.3e" % (amp, s_, r), flush=True)
    # linear response: ~13 x the noise at this depth; a mask flip of the exit flow shows as >= 1e-3
    assert worst[1e-6] <= 1e-4, "knife-edge: 1e-6 input noise moves the float64 gradient by %.2e" % worst[1e-6]
    assert worst[1e-5] <= 5e-4, "knife-edge: 1e-5 input noise moves the float64 gradient by %.2e" % worst[1e-5]
    print("OK: no ReLU input of this fixture sits within float32 rounding of zero "
          "(1e-6 -> %.2e, 1e-5 -> %.2e)" % (worst[1e-6], worst[1e-5]))


if __name__ == "__main__":
    main()
