#!/usr/bin/env python3
"""Fixtures for the other BASELINE configs, generated from the REFERENCE (dev container only).

    python oracle/gen_golden_more.py c1     # FCN resnet101 (OS16)           -> tests/golden/c1_*
    python oracle/gen_golden_more.py c4     # PSPNet resnet101 (OS8, aux)    -> tests/golden/c4_*
    python oracle/gen_golden_more.py c2     # DeepLabv3+ mobilenet_v2        -> tests/golden/c2_*
    python oracle/gen_golden_more.py c5     # HRNet hrnet_w18_small_v1       -> tests/golden/c5_*
    python oracle/gen_golden_more.py c6     # CCNet resnet101 (stubbed _C)   -> tests/golden/c6_*
    python oracle/gen_golden_more.py c9     # PSPNet resnet50, BN_TYPE GN    -> tests/golden/c9_*

One process per model (the reference cfg singleton freezes).  C1 / C4 use resnet101, the
backbone BASELINE.md names (BASELINE C1 as written, "FCN-resnet18", cannot run in the reference:
fcn.py:16 hard-codes 2048 input channels, F4).
Each run asserts that oracle/torch_ref.py reproduces the reference bit-for-bit (forward and
every parameter gradient) before writing:
  <tag>_state_keys.json, <tag>_bn_calib.npz, <tag>_eval.npz (logits), <tag>_train.npz
  (loss, per-output logits checksum, grad norms, a few full gradients).
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_import, synth, torch_ref  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
CASES = {
    "c1": dict(yaml="configs/cityscapes_fcn.yaml", over=["MODEL.BACKBONE", "resnet101"],
               fn="fcn_resnet", os=16, aux=False, hw=(65, 97), eps_enc=None),
    "c4": dict(yaml="configs/cityscapes_pspnet_resnet.yaml", over=["MODEL.BACKBONE", "resnet101"],
               fn="pspnet_resnet", os=8, aux=True, hw=(49, 65), eps_enc=None),
    # cfg.MODEL.BN_TYPE 'GN' (modules/batch_norm.py:105-108,129): GroupNorm(min(32, C), C) in the
    # encoder; the reference's PSPNet builds its heads without norm_layer (pspnet.py:23-25), so they
    # keep BatchNorm2d.  resnet50: every width divisible by 32
    "c9": dict(yaml="configs/cityscapes_pspnet_resnet.yaml",
               over=["MODEL.BACKBONE", "resnet50", "MODEL.BN_TYPE", "GN"],
               fn="pspnet_resnet", os=8, aux=True, hw=(49, 65), eps_enc=None, norm="GN"),
    "c2": dict(yaml="configs/cityscapes_deeplabv3_plus_mobilenet.yaml", over=[],
               fn="deeplab_mobilenet", os=16, aux=False, hw=(65, 97), eps_enc=None),
    # HRNet needs H, W divisible by 32 (nearest x2 upsamples must meet the stride-2 conv sizes)
    "c5": dict(yaml="configs/cityscapes_hrnet_w18_small_v1.yaml", over=[], fn="hrnet_seg", os=16,
               aux=False, hw=(64, 128), eps_enc=None, mom=0.01),
    # CCNet: the model is disabled in the reference (models/__init__.py:11) because its CUDA
    # extension `segmentron._C` is not built; a stub `_C` backed by oracle.torch_ref's restatement
    # of ca_cuda.cu lets the REFERENCE's own module tree / wiring / autograd glue run on the CPU
    "c6": dict(yaml="configs/cityscapes_ccnet_resnet.yaml", over=[], fn="ccnet_resnet", os=16,
               aux=False, hw=(65, 97), eps_enc=None, pre="ccnet"),
    # Fast-SCNN (SURVEY §8 f4 tail, README: 145.77 FPS on V100): no backbone, three outputs
    # (SOLVER.AUX True in its yaml).  Size: the x4 upsample of the 1/32 branch must meet the 1/8
    # branch and the pyramid pooling needs a >= 6x6 map -> 192x192 (6x6 at 1/32); logits are
    # stored every 2nd pixel (`sub`) to keep the fixture small
    # DANet (SURVEY §8 f4 tail): resnet101 OS8 with the multi-grid layer4 (dilations 4/8/16),
    # position + channel attention heads, three outputs
    "c8": dict(yaml="configs/cityscapes_danet_resnet.yaml", over=[], fn="danet_resnet", os=8,
               aux=False, hw=(49, 65), eps_enc=None, multi_dilation=[4, 8, 16], gamma=True),
    "c7": dict(yaml="configs/cityscapes_fast_scnn.yaml", over=["TEST.TEST_MODEL_PATH", ""], fn="fast_scnn", os=16, aux=True,
               hw=(192, 192), eps_enc=None, mom=0.01, sub=2),
}


def _install_ccnet():
    """segmentron._C stand-in (ca.h:25-73 entry points) + explicit import of models.ccnet."""
    import types
    from oracle import ref_import as ri
    ri._install_stubs()
    if ri.REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, ri.REFERENCE_ROOT)
    import segmentron
    assert segmentron.__file__.startswith(ri.REFERENCE_ROOT)
    C = types.ModuleType("segmentron._C")

    def _grads(fn, args, dout):
        leaves = [a.detach().requires_grad_(True) for a in args]
        with torch.enable_grad():
            out = fn(*leaves)
        return torch.autograd.grad(out, leaves, dout)

    C.ca_forward = lambda t, f: torch_ref.cca_weight(t, f)
    C.ca_backward = lambda dw, t, f: _grads(torch_ref.cca_weight, (t, f), dw)
    C.ca_map_forward = lambda w, g: torch_ref.cca_map(w, g)
    C.ca_map_backward = lambda dout, w, g: _grads(torch_ref.cca_map, (w, g), dout)
    sys.modules["segmentron._C"] = C
    segmentron._C = C
    import segmentron.models.ccnet  # noqa: F401  (registers CCNet in MODEL_REGISTRY)



def main(tag):
    c = CASES[tag]
    torch.set_num_threads(min(16, os.cpu_count()))
    if c.get("pre") == "ccnet":
        _install_ccnet()
    model, cfg = ref_import.build_reference_model(c["yaml"], c["over"])
    ref_import.apply_bn_attrs(model, cfg)
    sd0 = model.state_dict()
    json.dump({"model": cfg.MODEL.MODEL_NAME, "backbone": cfg.MODEL.BACKBONE, "config": c["yaml"],
               "overrides": c["over"], "n_params": int(sum(p.numel() for p in model.parameters())),
               "keys": [(k, list(v.shape)) for k, v in sd0.items()]},
              open(os.path.join(GOLD, tag + "_state_keys.json"), "w"))
    sd = synth.synth_like(sd0, seed=0)
    model.load_state_dict(sd, strict=True)
    H, W = c["hw"]
    x = synth.synth_images(2, H, W, seed=0)
    y = synth.synth_targets(2, H, W, seed=0)
    drops = [m for m in model.modules() if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d))]
    for d in drops:
        d.p = 0.0
    # calibrate BN running stats (see gen_golden.py)
    bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    saved = [m.momentum for m in bns]
    for m in bns:
        m.momentum = 1.0
    model.train()
    with torch.no_grad():
        model(x)
    for m, mom in zip(bns, saved):
        m.momentum = mom
    calib = {}
    msd = model.state_dict()
    for k, v in msd.items():
        if k.endswith("running_var"):
            calib[k] = (v * (0.8 + 0.45 * torch.rand(v.shape, generator=synth._gen(7, k)))).clone()
        elif k.endswith("running_mean"):
            rv = msd[k[:-4] + "var"]
            calib[k] = (v + 0.05 * rv.sqrt() * torch.randn(v.shape, generator=synth._gen(7, k))).clone()
    np.savez_compressed(os.path.join(GOLD, tag + "_bn_calib.npz"),
                        **{k: v.numpy() for k, v in calib.items()})
    sd.update(calib)
    for k in sd:
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.long)
    model.load_state_dict(sd, strict=True)

    kw = dict(output_stride=c["os"], aux=c["aux"], eps_encoder=c["eps_enc"], drop_p=0.0,
              momentum=c.get("mom"), multi_dilation=c.get("multi_dilation"),
              norm=c.get("norm", "BN"))
    model.eval()
    with torch.no_grad():
        outs = model(x)
        net = torch_ref.OracleNet(torch_ref.clone_state(sd), training=False, **kw)
        o_outs = getattr(net, c["fn"])(x)
    for a, b in zip(outs, o_outs):
        assert (a - b).abs().max().item() == 0.0, "oracle differs from the reference (eval)"
    print(tag, "eval logits", tuple(outs[0].shape), "absmax %.3f" % outs[0].abs().max().item())
    sub = c.get("sub", 1)
    np.savez_compressed(os.path.join(GOLD, tag + "_eval.npz"),
                        logits=outs[0][..., ::sub, ::sub].numpy(),
                        argmax=outs[0].argmax(1).to(torch.uint8)[..., ::sub, ::sub].numpy(),
                        sub=np.int64(sub))

    model.train()
    model.zero_grad()
    outs = model(x)
    loss = torch_ref.mix_softmax_ce(outs, y, aux_weight=cfg.SOLVER.AUX_WEIGHT)
    loss.backward()
    grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    osd = torch_ref.clone_state(sd, requires_grad=True)
    net = torch_ref.OracleNet(osd, training=True, **kw)
    o_outs = getattr(net, c["fn"])(x)
    o_loss = torch_ref.mix_softmax_ce(o_outs, y, aux_weight=cfg.SOLVER.AUX_WEIGHT)
    o_loss.backward()
    assert abs(loss.item() - o_loss.item()) == 0.0
    worst = max((osd[k].grad - g).abs().max().item() for k, g in grads.items())
    assert worst == 0.0, "oracle backward differs from the reference: %g" % worst
    print(tag, "train loss %.6f; oracle == reference bit-for-bit (fwd + %d grads)" % (loss.item(), len(grads)))
    # running statistics after the train step (momentum / unbiased-variance / conv-bias paths)
    msd, stat_err = model.state_dict(), 0.0
    for k in msd:
        if k.endswith("running_mean") or k.endswith("running_var"):
            stat_err = max(stat_err, (msd[k] - osd[k].detach()).abs().max().item())
    assert stat_err == 0.0, "oracle running stats differ from the reference: %g" % stat_err
    names = list(grads)
    stat_keys = [k for k in msd if k.endswith("running_mean")]
    stat_keys = stat_keys[:2] + stat_keys[-2:]
    stat_keys += [k[:-4] + "var" for k in stat_keys]
    payload = {"loss": np.float64(loss.item()),
               "logits": outs[0].detach()[..., ::sub, ::sub].numpy(), "sub": np.int64(sub),
               "grad_norm_keys": np.array(names),
               "grad_norms": np.array([float(grads[k].double().norm()) for k in names])}
    for k in names[:2] + names[len(names) // 2:len(names) // 2 + 2] + names[-2:]:
        if grads[k].numel() <= 300000:
            payload["grad::" + k] = grads[k].numpy()
    for k in stat_keys:
        payload["stat::" + k] = msd[k].numpy()
    np.savez_compressed(os.path.join(GOLD, tag + "_train.npz"), **payload)


if __name__ == "__main__":
    main(sys.argv[1])
