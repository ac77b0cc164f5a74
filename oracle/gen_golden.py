#!/usr/bin/env python3
"""Generate tests/golden/* from the REFERENCE itself (dev container only; TEST INFRASTRUCTURE).

    python oracle/gen_golden.py            # writes tests/golden/c3_*.npz + *.json

The reference (LikeLy-Journey/SegmenTron @ /root/reference) has no tests or golden vectors
(SURVEY.md §4), so the pin for the oracle is "outputs of the reference run here": the reference
model is built through its own config + registry path (oracle/ref_import.py), loaded with
``oracle.synth`` parameters, and run on ``oracle.synth`` inputs in fp32 on CPU.

Fixtures (C3 = DeepLabv3+ xception65, configs/cityscapes_deeplabv3_plus.yaml):
  c3_state_keys.json       every state_dict key + shape of the reference model (Appendix E)
  c3_bn_calib.npz          calibrated+perturbed BN running stats (load on top of synth weights)
  c3_eval_65x129.npz       eval-mode logits [2,19,65,129], BN eps as optimizer.py would set
  c3_train_65x129.npz      train-mode (dropout p=0): logits, loss, per-parameter grad L2 norms,
                           a few full gradients, updated BN running stats of selected layers
One process per model: the reference cfg singleton freezes.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_import, synth  # noqa: E402
from oracle import torch_ref  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
FULL_GRAD_KEYS = [
    "encoder.conv1.weight",
    "encoder.bn1.weight",
    "encoder.block1.sep_conv1.block.depthwise.weight",
    "encoder.block2.conv.weight",
    "encoder.block10.sep_conv2.block.bn_depth.bias",
    "encoder.block21.sep_conv3.block.bn_point.weight",
    "head.aspp.image_pooling.conv.weight",
    "head.aspp.aspp2.block.depthwise.weight",
    "head.c1_block.conv.weight",
    "head.block.2.weight",
    "head.block.2.bias",
]
STAT_KEYS = [
    "encoder.bn1", "encoder.block4.sep_conv1.block.bn_depth",
    "encoder.block21.sep_conv3.block.bn_point", "head.aspp.image_pooling.bn", "head.aspp.bn",
    "head.block.1.block.bn_point",
]


def main():
    torch.set_num_threads(os.cpu_count())
    os.makedirs(GOLD, exist_ok=True)
    model, cfg = ref_import.build_reference_model("configs/cityscapes_deeplabv3_plus.yaml")
    ref_import.apply_bn_attrs(model, cfg)
    sd0 = model.state_dict()
    keys = [(k, list(v.shape)) for k, v in sd0.items()]
    with open(os.path.join(GOLD, "c3_state_keys.json"), "w") as f:
        json.dump({"model": "DeepLabV3_Plus", "backbone": "xception65",
                   "config": "configs/cityscapes_deeplabv3_plus.yaml",
                   "n_params": int(sum(p.numel() for p in model.parameters())),
                   "keys": keys}, f)
    sd = synth.synth_like(sd0, seed=0)
    model.load_state_dict(sd, strict=True)
    B, H, W = 2, 65, 129
    x = synth.synth_images(B, H, W, seed=0)
    y = synth.synth_targets(B, H, W, seed=0)

    # ---- calibrate BN running stats: random running stats let activations explode through
    # 70 eval-mode layers (logits ~1e13), which would make an eval parity test meaningless.
    # One reference train-mode pass with momentum=1 makes running stats = batch stats; a
    # deterministic perturbation keeps eval-mode BN different from train-mode BN.
    bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    saved = [m.momentum for m in bns]
    for m in bns:
        m.momentum = 1.0
    model.train()
    model.head.aspp.dropout.p = 0.0
    with torch.no_grad():
        model(x)
    for m, mom in zip(bns, saved):
        m.momentum = mom
    calib = {}
    for k, v in model.state_dict().items():
        if k.endswith("running_var"):
            g = synth._gen(7, k)
            calib[k] = (v * (0.8 + 0.45 * torch.rand(v.shape, generator=g))).clone()
        elif k.endswith("running_mean"):
            g = synth._gen(7, k)
            rv = model.state_dict()[k[:-4] + "var"]
            calib[k] = (v + 0.05 * rv.sqrt() * torch.randn(v.shape, generator=g)).clone()
    np.savez_compressed(os.path.join(GOLD, "c3_bn_calib.npz"),
                        **{k: v.numpy() for k, v in calib.items()})
    sd.update(calib)
    for k in sd:
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.long)
    model.load_state_dict(sd, strict=True)

    # ---- eval
    model.eval()
    with torch.no_grad():
        logits = model(x)[0]
    orc = torch_ref.OracleNet(torch_ref.clone_state(sd), training=False, eps_encoder=1e-3)
    with torch.no_grad():
        o_logits = orc.deeplabv3_plus_xception65(x)[0]
    err = (o_logits - logits).abs().max().item()
    print("eval: ref logits", tuple(logits.shape), "absmax", logits.abs().max().item(),
          "oracle-vs-ref max abs", err)
    assert err == 0.0, "oracle restatement differs from the reference in eval mode"
    np.savez_compressed(os.path.join(GOLD, "c3_eval_65x129.npz"),
                        logits=logits.numpy(), argmax=logits.argmax(1).to(torch.uint8).numpy())

    # ---- train (dropout off: RNG-free parity; reference module p set to 0 in memory only)
    model.train()
    model.head.aspp.dropout.p = 0.0
    model.zero_grad()
    out = model(x)
    loss = torch.nn.functional.cross_entropy(out[0], y, ignore_index=-1)
    loss.backward()
    grads = {k: p.grad for k, p in model.named_parameters()}
    gn = {k: float(g.double().norm()) for k, g in grads.items()}
    sd_after = model.state_dict()

    osd = torch_ref.clone_state(sd, requires_grad=True)
    orc = torch_ref.OracleNet(osd, training=True, eps_encoder=1e-3, drop_p=0.0)
    o_out = orc.deeplabv3_plus_xception65(x)
    o_loss = torch_ref.mix_softmax_ce(o_out, y)
    o_loss.backward()
    print("train: loss ref %.9f oracle %.9f" % (loss.item(), o_loss.item()))
    assert (o_out[0] - out[0]).abs().max().item() == 0.0
    worst = max((osd[k].grad - g).abs().max().item() for k, g in grads.items())
    print("train: worst grad abs diff oracle-vs-ref", worst)
    assert worst == 0.0, "oracle backward differs from the reference"
    for k in STAT_KEYS:
        assert torch.equal(osd[k + ".running_var"], sd_after[k + ".running_var"])

    payload = {"logits": out[0].detach().numpy(), "loss": np.float64(loss.item()),
               "grad_norm_keys": np.array(list(gn.keys())),
               "grad_norms": np.array(list(gn.values()), dtype=np.float64)}
    for k in FULL_GRAD_KEYS:
        payload["grad::" + k] = grads[k].numpy()
    for k in STAT_KEYS:
        payload["rm::" + k] = sd_after[k + ".running_mean"].numpy()
        payload["rv::" + k] = sd_after[k + ".running_var"].numpy()
    np.savez_compressed(os.path.join(GOLD, "c3_train_65x129.npz"), **payload)
    print("wrote fixtures to", GOLD)


if __name__ == "__main__":
    main()
