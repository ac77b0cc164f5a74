#!/usr/bin/env python3
"""WELL-CONDITIONED end-to-end fixtures, generated from the REFERENCE (dev container only).

    python oracle/gen_golden_cond.py c3     # DeepLabv3+ xception65      -> tests/golden/c3_cond.npz
    python oracle/gen_golden_cond.py c2     # DeepLabv3+ mobilenet_v2    -> tests/golden/c2_cond.npz
    python oracle/gen_golden_cond.py c4     # PSPNet resnet101 (OS8,aux) -> tests/golden/c4_cond.npz
    python oracle/gen_golden_cond.py c5     # HRNet w18_small_v1         -> tests/golden/c5_cond.npz

TEST INFRASTRUCTURE.  Same recipe as gen_golden.py / gen_golden_more.py (reference model built
through its own cfg + registry, `oracle.synth` parameters and inputs, fp32 CPU), but with
`synth.synth_like(..., conditioned=True)`: the state on which end-to-end parity can be asserted
with FIXED bars, also for the bf16 throughput path.  Before anything is
written the script asserts
  (1) oracle/torch_ref.py == the reference bit-for-bit on this state (eval logits, train logits,
      loss, every parameter gradient, running statistics), and
  (2) the CONDITIONING: CPU fp32 vs fp64 oracle — logits L2-rel <= 1e-5, global gradient error
      <= 1.5e-4; conv weights rounded to bf16 (fp64 arithmetic) — logits L2-rel <= 1e-2, global
      gradient error <= 0.1.  (The default synth state gives 1.5e-4 / 1.7e-2 / 0.5 / 1.0 on C3.)
One process per model: the reference cfg singleton freezes.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_import, synth, torch_ref  # noqa: E402
from oracle.gen_golden_more import CASES as MORE  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
CASES = dict(MORE)
CASES["c3"] = dict(yaml="configs/cityscapes_deeplabv3_plus.yaml", over=[],
                   fn="deeplabv3_plus_xception65", os=16, aux=False, hw=(65, 129), eps_enc=1e-3)


def oracle_step(c, sd, x, y, dtype, aux_weight):
    s = {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    s = torch_ref.clone_state(s, requires_grad=True)
    net = torch_ref.OracleNet(s, training=True, output_stride=c["os"], aux=c["aux"],
                              eps_encoder=c["eps_enc"], drop_p=0.0, momentum=c.get("mom"))
    outs = getattr(net, c["fn"])(x.to(dtype))
    loss = torch_ref.mix_softmax_ce(outs, y, aux_weight=aux_weight)
    loss.backward()
    return outs, loss, {k: v.grad for k, v in s.items() if v.grad is not None}, s


def global_err(ga, gb):
    num = sum((ga[k].double() - t.double()).norm().item() ** 2 for k, t in gb.items())
    den = sum(t.double().norm().item() ** 2 for t in gb.values())
    return (num / den) ** 0.5


def l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


def main(tag):
    c = CASES[tag]
    torch.set_num_threads(min(16, os.cpu_count()))
    model, cfg = ref_import.build_reference_model(c["yaml"], c["over"])
    ref_import.apply_bn_attrs(model, cfg)
    sd = synth.synth_like(model.state_dict(), seed=0, conditioned=True)
    model.load_state_dict(sd, strict=True)
    H, W = c["hw"]
    x = synth.synth_images(2, H, W, seed=0)
    y = synth.synth_targets(2, H, W, seed=0)
    for d in model.modules():
        if isinstance(d, (torch.nn.Dropout, torch.nn.Dropout2d)):
            d.p = 0.0
    # calibrated + perturbed running statistics (see gen_golden.py)
    bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    saved = [m.momentum for m in bns]
    for m in bns:
        m.momentum = 1.0
    model.train()
    with torch.no_grad():
        model(x)
    for m, mom in zip(bns, saved):
        m.momentum = mom
    calib, msd = {}, model.state_dict()
    for k, v in msd.items():
        if k.endswith("running_var"):
            calib[k] = (v * (0.8 + 0.45 * torch.rand(v.shape, generator=synth._gen(7, k)))).clone()
        elif k.endswith("running_mean"):
            rv = msd[k[:-4] + "var"]
            calib[k] = (v + 0.05 * rv.sqrt() * torch.randn(v.shape, generator=synth._gen(7, k))).clone()
    sd.update(calib)
    for k in sd:
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.long)
    model.load_state_dict(sd, strict=True)
    kw = dict(output_stride=c["os"], aux=c["aux"], eps_encoder=c["eps_enc"], drop_p=0.0,
              momentum=c.get("mom"))

    # ---- eval: oracle == reference
    model.eval()
    with torch.no_grad():
        outs = model(x)
        net = torch_ref.OracleNet(torch_ref.clone_state(sd), training=False, **kw)
        o_outs = getattr(net, c["fn"])(x)
    for a, b in zip(outs, o_outs):
        assert (a - b).abs().max().item() == 0.0, "oracle differs from the reference (eval)"
    eval_logits = outs[0].clone()
    top2 = eval_logits.topk(2, dim=1).values
    print(tag, "eval logits absmax %.3f, smallest top-2 margin %.2e"
          % (eval_logits.abs().max().item(), (top2[:, 0] - top2[:, 1]).min().item()))

    # ---- train: oracle == reference (forward, loss, every gradient, running statistics)
    model.train()
    model.zero_grad()
    outs = model(x)
    aw = cfg.SOLVER.AUX_WEIGHT
    loss = torch_ref.mix_softmax_ce(outs, y, aux_weight=aw)
    loss.backward()
    grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    o32, l32, g32, s32 = oracle_step(c, sd, x, y, torch.float32, aw)
    assert loss.item() == l32.item() and (outs[0] - o32[0]).abs().max().item() == 0.0
    assert max((g32[k] - g).abs().max().item() for k, g in grads.items()) == 0.0, \
        "oracle backward differs from the reference"
    msd = {k: v.detach().clone() for k, v in model.state_dict().items()}  # (a snapshot: the
    #       autocast yardstick below runs another train step on the same module objects)
    assert all(torch.equal(msd[k], s32[k].detach()) for k in msd if "running_" in k)

    # ---- the conditioning itself
    o64, l64, g64, _ = oracle_step(c, sd, x, y, torch.float64, aw)
    sdb = {k: (v.to(torch.bfloat16).float() if (v.is_floating_point() and v.dim() == 4) else v)
           for k, v in sd.items()}
    ob, lb, gb, _ = oracle_step(c, sdb, x, y, torch.float64, aw)
    cond = dict(fp32_logits=l2(o32[0].detach(), o64[0].detach()), fp32_grads=global_err(g32, g64),
                bf16w_logits=l2(ob[0].detach(), o64[0].detach()), bf16w_grads=global_err(gb, g64))
    print(tag, "conditioning:", " ".join("%s %.2e" % kv for kv in cond.items()))
    assert cond["fp32_logits"] <= 1e-5 and cond["fp32_grads"] <= 1.5e-4, cond
    assert cond["bf16w_logits"] <= 1e-2 and cond["bf16w_grads"] <= 0.1, cond

    # ---- yardstick for the bf16 throughput path: the REFERENCE model itself under torch's CPU
    # bf16 autocast (what SURVEY F9 measured on the default init), on this state and input
    model.eval()
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        ac_eval = model(x)[0].float()
    model.train()
    model.load_state_dict(sd, strict=True)
    model.zero_grad()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        ac_outs = model(x)
    ac_loss = torch_ref.mix_softmax_ce([o.float() for o in ac_outs], y, aux_weight=aw)
    ac_loss.backward()
    dot = na = nb = 0.0
    for k, p in model.named_parameters():
        if p.grad is not None and k in g64:
            dot += (p.grad.double() * g64[k]).sum().item()
            na += p.grad.double().norm().item() ** 2
            nb += g64[k].norm().item() ** 2
    ref_eval64 = eval_logits
    autocast = dict(eval_l2=l2(ac_eval, ref_eval64),
                    eval_argmax=(ac_eval.argmax(1) == ref_eval64.argmax(1)).float().mean().item(),
                    train_l2=l2(ac_outs[0].detach().float(), o64[0].detach()),
                    loss_rel=abs(ac_loss.item() - l64.item()) / l64.item(),
                    grad_cos=dot / (na * nb) ** 0.5, grad_norm_ratio=(na / nb) ** 0.5)
    print(tag, "reference under CPU bf16 autocast:", " ".join("%s %.4g" % kv for kv in autocast.items()))

    names = list(grads)
    payload = {"eval_logits": eval_logits.numpy(),
               "train_logits": outs[0].detach().numpy(), "loss": np.float64(loss.item()),
               "loss64": np.float64(l64.item()),
               "grad_norm_keys": np.array(names),
               "grad_norms": np.array([float(grads[k].double().norm()) for k in names]),
               "grad_norms64": np.array([float(g64[k].norm()) for k in names]),
               "conditioning": np.array([cond[k] for k in ("fp32_logits", "fp32_grads",
                                                            "bf16w_logits", "bf16w_grads")]),
               # eval_l2, eval_argmax, train_l2, loss_rel, grad_cos, grad_norm_ratio
               "ref_autocast_bf16": np.array([autocast[k] for k in (
                   "eval_l2", "eval_argmax", "train_l2", "loss_rel", "grad_cos",
                   "grad_norm_ratio")])}
    for k, v in calib.items():
        payload["calib::" + k] = v.numpy()
    stat_keys = [k for k in msd if k.endswith("running_mean")]
    for k in stat_keys[:2] + stat_keys[-2:]:
        payload["stat::" + k] = msd[k].numpy()
        payload["stat::" + k[:-4] + "var"] = msd[k[:-4] + "var"].numpy()
    np.savez_compressed(os.path.join(GOLD, tag + "_cond.npz"), **payload)
    print(tag, "wrote", os.path.join(GOLD, tag + "_cond.npz"))


if __name__ == "__main__":
    main(sys.argv[1])
