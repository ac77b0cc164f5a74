"""End-to-end parity on the WELL-CONDITIONED fixtures (tests/golden/<tag>_cond.npz, generated from
the reference by oracle/gen_golden_cond.py; `oracle.synth(conditioned=True)`), with FIXED bars —
also for the bf16 throughput path that bench.py measures.

On the default synth state a random 70-layer ReLU+BatchNorm chain is a chaotic map (CPU fp32
gradients 1.7e-2 from fp64, bf16 gradients decorrelated), so the whole-model tests in
test_model_gpu.py / test_more_models.py can only bound the HIP error relative to the CPU's own.
Here the generator asserted the conditioning (CPU fp32 vs fp64: logits <= 1e-5, gradients
<= 1.5e-4 global) before writing, so the bars below are absolute:

  fp32 path   eval / train logits max-rel <= 1e-3 of the reference fixture, arg-max masks identical
              (up to oracle top-2 ties, counted and printed), loss 1e-3, running statistics 1e-3,
              gradients <= 1e-3 global-rel of the fp64 oracle, every tensor <= 1e-1 of its norm
              + 1e-4 of the global norm (a single ReLU / ReLU6 mask flip moves a small tensor by
              1/sqrt(its size): C2 measured 4.5e-2 on one 384-element BatchNorm weight);
  bf16 path   vs the fp32 fixture / fp64 oracle, yardstick = the REFERENCE ITSELF under torch's
              CPU bf16 autocast on the same state and input (stored in the fixture:
              `ref_autocast_bf16`): logits L2-rel <= max(2e-2, 1.5 x reference-autocast), arg-max
              agreement >= reference-autocast - 0.03, loss 1e-2, global gradient cosine >=
              min(0.99, reference-autocast) - 0.03, norm ratio within max(10 %, 1.5 x the
              reference-autocast's own deviation);
  (no "tight" bf16 check against oracle/bf16_emulation.py exists on purpose: measured r04 — two bf16
              pipelines with the SAME rounding points that differ only in fp32 accumulation order
              (1e-7) decorrelate to the bf16 noise floor within ~5 layers, because a difference
              delta << ulp moves a fraction delta/ulp of the elements across a rounding boundary:
              delta' = sqrt(delta * ulp); per-BatchNorm statistics HIP vs emulation: 1e-8, 1e-6,
              1e-4, 1e-3, 2e-3 = the emulation's own distance to fp32.  The emulation stays the
              per-op / teacher-forced-composite reference, tests/test_composites_gpu.py.)
  graph       the same C3 bf16 step through GraphedTrainStep + FusedSGD, one eager step + 3 replays,
              against 4 SGD steps of the fp64 oracle (loss 5.30 -> 4.54 -> 4.17 -> 3.87): losses 2e-2,
              weight-update cosine >= 0.85, norm ratio within 15 % — the bf16 gradient's own cosine to fp64 is ~0.90 on this
              net (reference autocast: 0.92) — a stale weight pack or a gradient that is not
              re-accumulated inside the graph leaves the loss flat and fails this.
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import C3_OVERRIDES, GOLDEN
from oracle import synth, torch_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {
    "c3": dict(fn="deeplabv3_plus_xception65", hw=(65, 129), os=16, aux=False, eps_enc=1e-3),
    "c2": dict(model="DeepLabV3_Plus", backbone="mobilenet_v2", os=16, aux=False,
               fn="deeplab_mobilenet", hw=(65, 97),
               over=["MODEL.DEEPLABV3_PLUS.USE_ASPP", "False",
                     "MODEL.DEEPLABV3_PLUS.ENABLE_DECODER", "False"]),
    "c4": dict(model="PSPNet", backbone="resnet101", os=8, aux=True, fn="pspnet_resnet",
               hw=(49, 65)),
    "c5": dict(model="HRNet", backbone="hrnet_w18_small_v1", os=16, aux=False, fn="hrnet_seg",
               hw=(64, 128), momentum=0.01, yaml="configs/cityscapes_hrnet_w18_small_v1.yaml"),
}
AUX_WEIGHT = 0.4
# what the bf16 path measured on the conditioned fixtures at r06 (gpurun_out r06b, printed as
# PARITY-COND ... bf16 by this file): eval / train logits L2-rel vs the fp32 fixture, arg-max
# agreement, gradient cosine / global-rel vs the fp64 oracle.  Bars = 1.3 x these.
BF16_MEASURED = {
    "c3": dict(eval_l2=3.995e-2, train_l2=1.335e-2, argmax=0.9664, cos=0.99375, grad_rel=1.127e-1),
    "c2": dict(eval_l2=8.493e-3, train_l2=6.913e-3, argmax=0.9910, cos=0.99993, grad_rel=1.250e-2),
    "c4": dict(eval_l2=3.028e-2, train_l2=1.845e-2, argmax=0.9692, cos=0.99151, grad_rel=1.308e-1),
    "c5": dict(eval_l2=1.382e-2, train_l2=1.196e-2, argmax=0.9771, cos=0.99993, grad_rel=1.182e-2),
}
# ... and at the large sizes (C3 513x1025, C3 / C4 1025x2049): logits L2-rel, 1 - arg-max, cosine
BF16_MEASURED_LARGE = {
    "c3_513x1025": dict(logits_l2=1.432e-2, argmax=0.9881, cos=0.99703, grad_rel=7.736e-2),
    "c3_1025x2049": dict(logits_l2=1.428e-2, argmax=0.9857, cos=0.99995, loss_rel=7.74e-4),
    "c4_1025x2049": dict(logits_l2=1.364e-2, argmax=0.9950, cos=0.99993, loss_rel=8.79e-5),
    # C3 + auxiliary head at 65 x 129, seed 3 (9 x 17 maps: the noisiest geometry of the suite)
    "c3_aux_65x129": dict(logits_l2=1.796e-2, cos=0.97958),
    # PSPNet / resnet50 with GroupNorm in the encoder, 65 x 97, seed 5
    "pspnet_gn_65x97": dict(logits_l2=4.303e-2, cos=0.99484),
}


def _fixture(tag):
    return np.load(os.path.join(GOLDEN, tag + "_cond.npz"))


def _state(tag):
    keys = json.load(open(os.path.join(GOLDEN, tag + "_state_keys.json")))["keys"]
    sd = synth.synth_state_dict([(k, tuple(s)) for k, s in keys], seed=0, conditioned=True)
    g = _fixture(tag)
    for k in g.files:
        if k.startswith("calib::"):
            sd[k[7:]] = torch.from_numpy(g[k])
    for k in sd:
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.long)
    return sd


def _oracle(tag, sd, x, y=None, dtype=torch.float32, training=True):
    c = CASES[tag]
    s = {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    s = torch_ref.clone_state(s, requires_grad=training)
    net = torch_ref.OracleNet(s, training=training, output_stride=c["os"], aux=c["aux"],
                              eps_encoder=c.get("eps_enc"), drop_p=0.0, momentum=c.get("momentum"))
    outs = getattr(net, c["fn"])(x.to(dtype))
    if not training:
        return outs, None, None, s
    loss = torch_ref.mix_softmax_ce(outs, y, aux_weight=AUX_WEIGHT)
    loss.backward()
    return outs, loss.item(), {k: v.grad for k, v in s.items() if v.grad is not None}, s


def _autocast_yardstick(size):
    """The oracle under torch's CPU bf16 autocast against its fp32 run at a LARGE size
    (tests/golden/c3_autocast_sizes.json, generated by oracle/gen_autocast_sizes.py): the arg-max
    agreement / logits distance of two bf16 pipelines on near-tie random-init logits, and the
    gradient cosine — dominated by the 2-sample BatchNorm of the image pooling branch — do not
    carry over from the 65x129 fixture."""
    return json.load(open(os.path.join(GOLDEN, "c3_autocast_sizes.json")))[size]


def _l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


def _maxrel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max()).item()


# ----------------------------------------------------------------------------- CPU: the fixture
@pytest.mark.parametrize("tag", list(CASES))
def test_conditioned_fixture_is_reproduced_by_the_oracle_and_is_conditioned(tag):
    g = _fixture(tag)
    c3, c4, c5, c6 = g["conditioning"]
    assert c3 <= 1e-5 and c4 <= 1.5e-4 and c5 <= 1e-2 and c6 <= 0.1, g["conditioning"]
    sd = _state(tag)
    H, W = CASES[tag]["hw"]
    x = synth.synth_images(2, H, W, seed=0)
    y = synth.synth_targets(2, H, W, seed=0)
    with torch.no_grad():
        outs = _oracle(tag, sd, x, training=False)[0]
    assert torch.allclose(outs[0], torch.from_numpy(g["eval_logits"]), rtol=1e-4, atol=1e-4)
    outs, loss, grads, _ = _oracle(tag, sd, x, y)
    assert abs(loss - float(g["loss"])) < 1e-5
    assert torch.allclose(outs[0].detach(), torch.from_numpy(g["train_logits"]), rtol=1e-4, atol=1e-4)
    for k, n in zip([str(k) for k in g["grad_norm_keys"]], g["grad_norms"]):
        assert abs(float(grads[k].double().norm()) - n) <= 1e-3 * max(n, 1e-6) + 1e-9, k


def test_conditioned_state_only_touches_batchnorm_affine_parameters():
    keys = json.load(open(os.path.join(GOLDEN, "c3_state_keys.json")))["keys"]
    a = synth.synth_state_dict([(k, tuple(s)) for k, s in keys], seed=0)
    b = synth.synth_state_dict([(k, tuple(s)) for k, s in keys], seed=0, conditioned=True)
    changed = [k for k in a if not torch.equal(a[k], b[k])]
    assert changed and all((k[:k.rfind(".")] + ".running_mean") in a for k in changed)
    last = [k for k in changed if synth.branch_last_bn(k, set(a))]
    assert len(last) == 20 and all(k.endswith("sep_conv3.block.bn_point.weight") for k in last)
    assert all(0.1 <= b[k].min() and b[k].max() <= 0.2 for k in last)


# ----------------------------------------------------------------------------- GPU
def _build_hip(tag, dtype, train):
    import segmentron_amd
    from segmentron_amd.config import cfg, reset_cfg
    c = CASES[tag]
    reset_cfg()
    if tag == "c3":
        cfg.update_from_list(C3_OVERRIDES)
    else:
        if "yaml" in c:
            cfg.update_from_file(os.path.join(ROOT, c["yaml"]))
        cfg.update_from_list(["DATASET.NAME", "cityscape", "MODEL.MODEL_NAME", c["model"],
                              "MODEL.BACKBONE", c["backbone"], "MODEL.OUTPUT_STRIDE", str(c["os"]),
                              "SOLVER.AUX", str(c["aux"]), "SOLVER.AUX_WEIGHT", str(AUX_WEIGHT),
                              "TRAIN.BACKBONE_PRETRAINED", "False"] + c.get("over", []))
    cfg.PHASE = "test"
    cfg.check_and_freeze()
    segmentron_amd.set_compute_dtype(dtype)
    model = segmentron_amd.get_segmentation_model()
    sd = _state(tag)
    model.load_state_dict(sd, strict=True)
    if tag == "c3":  # solver/optimizer.py:18-20
        for _, m in model.encoder.named_modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.eps = 1e-3
    model = model.cuda().train(train)
    for m in model.modules():
        if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)):
            m.p = 0.0
        if isinstance(m, torch.nn.BatchNorm2d) and c.get("momentum") is not None:
            m.momentum = c["momentum"]
    return model, sd


def _hip_step(model, x, y):
    ce = torch.nn.functional.cross_entropy
    outs = model(x.cuda())
    loss = ce(outs[0], y.cuda(), ignore_index=-1)
    for o in outs[1:]:
        loss = loss + AUX_WEIGHT * ce(o, y.cuda(), ignore_index=-1)
    loss.backward()
    return outs, loss


def _grad_stats(params, ref):
    """-> dict(global_rel, cos, ratio), [(rel, cos, ratio, share, key)] per tensor."""
    num = den = dot = na = 0.0
    per = []
    for k, t in ref.items():
        assert params[k].grad is not None, k
        a = params[k].grad.detach().cpu().double()
        assert torch.isfinite(a).all(), k
        t = t.double()
        e, n, m = (a - t).norm().item(), t.norm().item(), a.norm().item()
        d = (a * t).sum().item()
        num, den, dot, na = num + e * e, den + n * n, dot + d, na + m * m
        per.append([e, d / max(m * n, 1e-300), m / max(n, 1e-300), n, k])
    for p in per:
        p[3] = p[3] ** 2 / den
    return dict(global_rel=(num / den) ** 0.5, cos=dot / (na * den) ** 0.5,
                ratio=(na / den) ** 0.5, norm=den ** 0.5), per


def _count_ties(got, ref, what):
    """arg-max masks identical except where the reference's own top-2 margin is below the 1e-3
    bar of the logits themselves; returns the number of such tie pixels (printed by callers)."""
    bad = got.argmax(1) != ref.argmax(1)
    n = int(bad.sum())
    if n:
        top2 = ref.topk(2, dim=1).values
        gap = (top2[:, 0] - top2[:, 1])[bad]
        assert gap.max().item() < 1e-3 * ref.abs().max().item(), (what, n, gap.max().item())
    return n


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(CASES))
def test_fp32_eval_and_train_step_fixed_bars(tag):
    g = _fixture(tag)
    H, W = CASES[tag]["hw"]
    x = synth.synth_images(2, H, W, seed=0)
    y = synth.synth_targets(2, H, W, seed=0)
    model, sd = _build_hip(tag, torch.float32, False)
    with torch.no_grad():
        got = model(x.cuda())[0].cpu()
    ref = torch.from_numpy(g["eval_logits"])
    rel = _maxrel(got, ref)
    ties = _count_ties(got, ref, tag + " eval")
    assert rel < 1e-3
    model.train()
    outs, loss = _hip_step(model, x, y)
    rel_t = _maxrel(outs[0].detach().cpu(), torch.from_numpy(g["train_logits"]))
    assert abs(loss.item() - float(g["loss"])) < 1e-3 * float(g["loss"]) and rel_t < 1e-3
    msd = model.state_dict()
    for k in g.files:
        if k.startswith("stat::"):
            r = torch.from_numpy(g[k])
            assert (msd[k[6:]].cpu() - r).abs().max().item() <= 1e-3 * r.abs().max().item() + 1e-6, k
    _, l64, g64, _ = _oracle(tag, sd, x, y, torch.float64)
    st, per = _grad_stats(dict(model.named_parameters()), g64)
    worst = max(per, key=lambda p: p[0] / (1e-1 * (p[3] ** 0.5) * st["norm"] + 1e-4 * st["norm"]))
    print("PARITY-COND %s fp32: eval max-rel %.2e (%d tie pixels), train logits %.2e, loss %.6f vs "
          "%.6f | gradients vs fp64 oracle: global rel %.2e cosine %.6f; worst tensor %s rel %.2e"
          % (tag, rel, ties, rel_t, loss.item(), float(g["loss"]), st["global_rel"], st["cos"],
             worst[4], worst[0] / max(worst[3] ** 0.5 * st["norm"], 1e-300)))
    assert st["global_rel"] <= 1e-3
    for e, cos, ratio, share, k in per:
        assert e <= 1e-1 * share ** 0.5 * st["norm"] + 1e-4 * st["norm"], (k, e, share)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(CASES))
def test_bf16_eval_and_train_step_vs_oracle_with_the_reference_autocast_yardstick(tag):
    g = _fixture(tag)
    ac_eval, ac_argmax, ac_train, ac_loss, ac_cos, ac_ratio = g["ref_autocast_bf16"]
    H, W = CASES[tag]["hw"]
    x = synth.synth_images(2, H, W, seed=0)
    y = synth.synth_targets(2, H, W, seed=0)
    model, sd = _build_hip(tag, torch.bfloat16, False)
    with torch.no_grad():
        got = model(x.cuda())[0].float().cpu()
    ref = torch.from_numpy(g["eval_logits"])
    l2e = _l2(got, ref)
    agree = (got.argmax(1) == ref.argmax(1)).float().mean().item()
    model.train()
    outs, loss = _hip_step(model, x, y)
    _, l64, g64, _ = _oracle(tag, sd, x, y, torch.float64)
    l2t = _l2(outs[0].detach().float().cpu(), torch.from_numpy(g["train_logits"]))
    st, per = _grad_stats(dict(model.named_parameters()), g64)
    print("PARITY-COND %s bf16 vs fp32 fixture / fp64 oracle: eval logits L2-rel %.3e argmax %.4f | "
          "train logits %.3e loss %.5f vs %.5f | gradients global rel %.3e cosine %.5f norm ratio "
          "%.4f || reference under CPU bf16 autocast: eval %.3e argmax %.4f train %.3e cosine %.5f"
          % (tag, l2e, agree, l2t, loss.item(), l64, st["global_rel"], st["cos"], st["ratio"],
             ac_eval, ac_argmax, ac_train, ac_cos))
    assert l2e <= max(2e-2, 1.5 * ac_eval) and l2t <= max(2e-2, 1.5 * ac_train)
    assert agree >= ac_argmax - 0.03
    assert abs(loss.item() - l64) <= 1e-2 * l64
    assert st["cos"] >= min(0.99, ac_cos) - 0.03
    assert abs(st["ratio"] - 1.0) <= max(0.10, 1.5 * abs(ac_ratio - 1.0))
    # r06 (VERDICT r05 weak #1c): the yardstick above stopped guarding anything once the image-pooling
    # branch went to float32 (C3 cosine 0.8975 -> 0.994, bar 0.894) — hold the path to 1.3 x what it
    # MEASURES now (errors, 1 - cosine, 1 - arg-max agreement), per configuration
    m = BF16_MEASURED[tag]
    assert l2e <= 1.3 * m["eval_l2"] and l2t <= 1.3 * m["train_l2"], (l2e, l2t, m)
    assert 1.0 - agree <= 1.3 * (1.0 - m["argmax"]) + 2e-3, (agree, m)
    assert 1.0 - st["cos"] <= 1.3 * (1.0 - m["cos"]) + 2e-5, (st["cos"], m)
    assert st["global_rel"] <= 1.3 * m["grad_rel"], (st["global_rel"], m)


@pytest.mark.gpu
def test_c3_fp32_and_bf16_train_step_513x1025_fixed_bars():
    """The smallest C3 geometry whose middle flow (2 x 33 x 65 = 4290 pixels) runs the 256-wide
    direct-to-LDS GEMMs in forward, data gradient and split weight gradient."""
    H, W = 513, 1025
    x = synth.synth_images(2, H, W, seed=12)
    y = synth.synth_targets(2, H, W, seed=12)
    sd = _state("c3")
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    o64, l64, g64, _ = _oracle("c3", sd, x, y, torch.float64)
    ref = o64[0].detach()
    model, _ = _build_hip("c3", torch.float32, True)
    outs, loss = _hip_step(model, x, y)
    rel = _maxrel(outs[0].detach().cpu(), ref)
    st, per = _grad_stats(dict(model.named_parameters()), g64)
    print("PARITY-COND c3 fp32 513x1025: logits max-rel %.2e loss %.6f vs %.6f | gradients vs fp64 "
          "oracle: global rel %.2e cosine %.7f" % (rel, loss.item(), l64, st["global_rel"], st["cos"]))
    # where the fp32 distance sits (share of the squared global error; the CPU oracle's own fp32
    # run is 4.1e-4 from float64 at this size — tools/tmp measurement r05)
    top = sorted(per, key=lambda p: -p[0])[:8]
    tot = sum(p[0] ** 2 for p in per)
    print("PARITY-COND c3 fp32 513x1025 error shares: " + "; ".join(
        "%s %.0f%% (rel %.1e)" % (k, 100 * e * e / tot, e / max(share ** 0.5 * st["norm"], 1e-300))
        for e, cos, ratio, share, k in top))
    assert rel < 1e-3 and abs(loss.item() - l64) < 1e-3 * l64
    assert st["global_rel"] <= 1e-3
    del model, outs, loss
    model, _ = _build_hip("c3", torch.bfloat16, True)
    outs, loss = _hip_step(model, x, y)
    l2t = _l2(outs[0].detach().float().cpu(), ref)
    agree = (outs[0].detach().cpu().argmax(1) == ref.argmax(1)).float().mean().item()
    st, per = _grad_stats(dict(model.named_parameters()), g64)
    ac = _autocast_yardstick("513x1025")
    print("PARITY-COND c3 bf16 513x1025 vs fp64 oracle: logits L2-rel %.3e argmax %.4f loss %.5f vs "
          "%.5f | gradients global rel %.3e cosine %.5f ratio %.4f || the oracle under CPU bf16 "
          "autocast at this size: logits %.3e argmax %.4f cosine %.5f ratio %.4f"
          % (l2t, agree, loss.item(), l64, st["global_rel"], st["cos"], st["ratio"],
             ac["logits_l2rel"], ac["argmax_agree"], ac["grad_cosine"], ac["grad_norm_ratio"]))
    assert l2t <= max(2e-2, 1.5 * ac["logits_l2rel"]) and agree >= ac["argmax_agree"] - 0.03
    assert abs(loss.item() - l64) <= 1e-2 * l64
    assert st["cos"] >= min(0.99, ac["grad_cosine"]) - 0.03
    assert abs(st["ratio"] - 1.0) <= max(0.10, 1.5 * abs(ac["grad_norm_ratio"] - 1.0))
    m = BF16_MEASURED_LARGE["c3_513x1025"]  # (1.3 x measured, see BF16_MEASURED)
    assert l2t <= 1.3 * m["logits_l2"] and 1.0 - agree <= 1.3 * (1.0 - m["argmax"]) + 2e-3
    assert 1.0 - st["cos"] <= 1.3 * (1.0 - m["cos"]) + 2e-5 and st["global_rel"] <= 1.3 * m["grad_rel"]


@pytest.mark.gpu
def test_c3_train_full_size_1025x2049_matches_oracle():
    """The configuration bench.py is quoted on (BASELINE.json configs[2]: train, batch 2
    @1025x2049), one step, against the CPU fp32 oracle's step on the same conditioned state —
    the comparison bench.py emits as `parity` (oracle/parity.py; VERDICT r04 Missing #1;
    /root/reference/tools/train.py:135-146).  fp32 kernels: north_star's 1e-3 on loss, logits
    (a [::16, ::16] pixel grid) and the global gradient; bf16: the reference-under-autocast
    yardstick of the fixture, as in the smaller tests above."""
    from oracle import parity as OP
    H, W = 1025, 2049
    sd = _state("c3")
    x, y = OP.inputs(2, H, W, seed=0)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ref = OP.oracle_step(sd, x, y, torch.float32)
    _build_hip("c3", torch.float32, True)  # (sets cfg; OP.hip_step builds its own models)
    f32 = OP.compare(OP.hip_step("fp32", sd, x, y), ref)
    b16 = OP.compare(OP.hip_step("bf16", sd, x, y), ref)
    ac = _autocast_yardstick("1025x2049")
    print("PARITY-COND c3 FULL SIZE 1025x2049 B=2 vs CPU fp32 oracle (%.0f s): fp32 loss rel %.2e "
          "logits max-rel %.2e gradients global rel %.2e cosine %.7f | bf16 loss rel %.2e logits "
          "L2-rel %.3e argmax %.4f gradients cosine %.5f ratio %.4f || the oracle under CPU bf16 "
          "autocast at this size: logits %.3e argmax %.4f cosine %.5f ratio %.4f"
          % (ref["seconds"], f32["loss_rel"], f32["logits_maxrel"], f32["grad_global_rel"],
             f32["grad_cosine"], b16["loss_rel"], b16["logits_l2rel"], b16["argmax_agree"],
             b16["grad_cosine"], b16["grad_norm_ratio"], ac["logits_l2rel"], ac["argmax_agree"],
             ac["grad_cosine"], ac["grad_norm_ratio"]))
    for tag, c in (("fp32", f32), ("bf16", b16)):
        print("PARITY-COND c3 FULL SIZE %s gradient error shares: %s" % (tag, "; ".join(
            "%s %.0f%% (cosine %.3f)" % (t["tensor"], 100 * t["share_of_sq_error"], t["cosine"])
            for t in c["grad_error_top"][:5])))
    assert f32["finite"] and b16["finite"]
    assert f32["grad_tensors_missing"] == 0 and b16["grad_tensors_missing"] == 0
    assert f32["loss_rel"] < 1e-3 and f32["logits_maxrel"] < 1e-3
    assert f32["grad_global_rel"] <= FULL_SIZE_GRAD_BAR_FP32
    _assert_fp32_argmax(f32, "c3 FULL SIZE")
    assert b16["loss_rel"] <= 1e-2
    assert b16["logits_l2rel"] <= max(2e-2, 1.5 * ac["logits_l2rel"])
    assert b16["argmax_agree"] >= ac["argmax_agree"] - 0.03
    assert b16["grad_cosine"] >= min(0.99, ac["grad_cosine"]) - 0.03
    assert abs(b16["grad_norm_ratio"] - 1.0) <= max(0.10, 1.5 * abs(ac["grad_norm_ratio"] - 1.0))
    _assert_bf16_large(b16, "c3_1025x2049")


def _assert_fp32_argmax(cmp, what):
    """north_star: "argmax masks bit-identical" — every pixel where the fp32 path's arg-max differs
    from the oracle's must be one the oracle itself separates by less than the 1e-3 logits bar
    (oracle/parity.py::compare), and there may only be a handful of them."""
    print("PARITY-COND %s fp32 arg-max: %d of %d sampled pixels differ, %d of them NOT oracle "
          "top-2 near-ties (largest margin %.2e)"
          % (what, cmp["argmax_mismatch"], cmp["argmax_pixels"], cmp["argmax_unexplained"],
             cmp["argmax_largest_margin"]))
    assert cmp["argmax_unexplained"] == 0
    assert cmp["argmax_mismatch"] <= max(3, cmp["argmax_pixels"] // 2000)


def _assert_bf16_large(b16, key):
    m = BF16_MEASURED_LARGE[key]
    assert b16["logits_l2rel"] <= 1.3 * m["logits_l2"], (b16["logits_l2rel"], m)
    assert 1.0 - b16["argmax_agree"] <= 1.3 * (1.0 - m["argmax"]) + 2e-3, (b16["argmax_agree"], m)
    assert 1.0 - b16["grad_cosine"] <= 1.3 * (1.0 - m["cos"]) + 2e-5, (b16["grad_cosine"], m)
    if "loss_rel" in m:
        assert b16["loss_rel"] <= max(1e-3, 1.3 * m["loss_rel"]), (b16["loss_rel"], m)


@pytest.mark.gpu
def test_c4_train_full_size_1025x2049_matches_oracle():
    """BASELINE.json configs[3] at its OWN size: PSPNet-resnet101 (output stride 8, auxiliary
    head, SOLVER.AUX_WEIGHT 0.4) train step, batch 2 @1025x2049, against the CPU fp32 oracle's
    step on the same conditioned state (VERDICT r05 Missing #1;
    /root/reference/segmentron/models/pspnet.py:13-58, /root/reference/tools/train.py:135-146).
    fp32 kernels: 1e-3 on loss, logits and the global gradient, arg-max identical up to oracle
    near-ties; bf16: 1.3 x what the path measured when the test was written."""
    from oracle import parity as OP
    H, W = 1025, 2049
    sd = _state("c4")
    x, y = OP.inputs(2, H, W, seed=0)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ref = OP.oracle_step(sd, x, y, torch.float32, oracle_fn="pspnet_resnet", output_stride=8,
                         aux=True, eps_encoder=None)
    _build_hip("c4", torch.float32, True)  # (sets cfg; OP.hip_step builds its own models)
    f32 = OP.compare(OP.hip_step("fp32", sd, x, y, eps_encoder=None), ref)
    b16 = OP.compare(OP.hip_step("bf16", sd, x, y, eps_encoder=None), ref)
    print("PARITY-COND c4 FULL SIZE 1025x2049 B=2 vs CPU fp32 oracle (%.0f s): fp32 loss rel %.2e "
          "logits max-rel %.2e gradients global rel %.2e cosine %.7f | bf16 loss rel %.2e logits "
          "L2-rel %.3e argmax %.4f gradients global rel %.3e cosine %.5f ratio %.4f"
          % (ref["seconds"], f32["loss_rel"], f32["logits_maxrel"], f32["grad_global_rel"],
             f32["grad_cosine"], b16["loss_rel"], b16["logits_l2rel"], b16["argmax_agree"],
             b16["grad_global_rel"], b16["grad_cosine"], b16["grad_norm_ratio"]))
    for tag, c in (("fp32", f32), ("bf16", b16)):
        print("PARITY-COND c4 FULL SIZE %s gradient error shares: %s" % (tag, "; ".join(
            "%s %.0f%% (cosine %.3f)" % (t["tensor"], 100 * t["share_of_sq_error"], t["cosine"])
            for t in c["grad_error_top"][:5])))
    assert f32["finite"] and b16["finite"]
    assert f32["grad_tensors_missing"] == 0 and b16["grad_tensors_missing"] == 0
    assert f32["loss_rel"] < 1e-3 and f32["logits_maxrel"] < 1e-3
    assert f32["grad_global_rel"] <= FULL_SIZE_GRAD_BAR_FP32
    _assert_fp32_argmax(f32, "c4 FULL SIZE")
    assert abs(b16["grad_norm_ratio"] - 1.0) <= 0.10
    _assert_bf16_large(b16, "c4_1025x2049")


# north_star's 1e-3 (against the fp32 CPU oracle, which itself sits FULL_SIZE_ORACLE_FP32_VS_FP64
# from float64 at this size — measured on the CPU)
FULL_SIZE_GRAD_BAR_FP32 = 1e-3


def _oracle_sgd_steps(sd, x, y, steps, lr, momentum, wd):
    """`steps` iterations of tools/train.py:135-146 in float64 on the oracle (torch.optim.SGD
    semantics: g += wd*w; buf = g | momentum*buf + g; w -= lr*buf)."""
    state = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    bufs, losses = {}, []
    for _ in range(steps):
        outs, loss, grads, s = _oracle("c3", state, x, y, torch.float64)
        losses.append(loss)
        nxt = {k: v.detach().clone() for k, v in s.items()}  # running statistics were updated
        for k, gk in grads.items():
            gk = gk + wd * s[k].detach()
            bufs[k] = gk.clone() if k not in bufs else momentum * bufs[k] + gk
            nxt[k] = s[k].detach() - lr * bufs[k]
        state = nxt
    return losses, state


@pytest.mark.gpu
def test_c3_bf16_graphed_train_steps_follow_the_oracle_trajectory():
    """GraphedTrainStep (forward + loss + backward + FusedSGD in ONE HIP graph), three replays,
    against three float64 SGD steps of the oracle from the same state.  lr is large enough that
    the loss moves by several percent per step: a weight pack that is not refreshed inside the
    graph, or a gradient that is not re-accumulated, leaves the loss flat."""
    from segmentron_amd.graph import GraphedTrainStep
    from segmentron_amd.solver.optimizer import FusedSGD
    H, W = 65, 129
    lr, mom, wd = 0.005, 0.9, 1e-4  # fp64 oracle: loss 5.30 -> 4.54 -> 4.17 -> 3.87
    x = synth.synth_images(2, H, W, seed=0)
    y = synth.synth_targets(2, H, W, seed=0)
    model, sd = _build_hip("c3", torch.bfloat16, True)
    w0 = {k: p.detach().cpu().double().clone() for k, p in model.named_parameters()}
    opt = FusedSGD(model.parameters(), lr=lr, momentum=mom, weight_decay=wd)
    ce = torch.nn.functional.cross_entropy
    # warmup=1: ONE real eager step first (it creates the momentum buffers, which must exist
    # before the capture); the capture itself does not execute, every replay is one real step
    step = GraphedTrainStep(model, opt, x.cuda(), y.cuda(),
                            lambda out, t: ce(out[0], t, ignore_index=-1), warmup=1)
    got = [step().item() for _ in range(3)]
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    want, state = _oracle_sgd_steps(sd, x, y, 4, lr, mom, wd)
    want = want[1:]
    dot = na = nb = 0.0
    for k, p in model.named_parameters():
        a = p.detach().cpu().double() - w0[k]
        b = state[k] - w0[k]
        dot, na, nb = dot + (a * b).sum().item(), na + a.norm().item() ** 2, nb + b.norm().item() ** 2
    cos = dot / (na * nb) ** 0.5
    print("PARITY-COND c3 bf16 graphed steps: losses %s vs fp64 oracle %s; weight-update cosine %.5f "
          "norm ratio %.4f" % (["%.4f" % v for v in got], ["%.4f" % v for v in want], cos,
                               (na / nb) ** 0.5))
    assert abs(want[0] - want[2]) > 0.03 * want[0], "lr too small for the check to bite"
    for a, b in zip(got, want):
        assert abs(a - b) <= 2e-2 * b, (got, want)
    # (the bf16 gradient itself: cosine 0.90 / norm ratio 1.12 vs fp64 on this net, reference
    # autocast 0.92 / 1.10 — test above; measured here 0.90-0.92 / 1.09-1.12)
    assert cos >= 0.85 and abs((na / nb) ** 0.5 - 1.0) <= 0.15


@pytest.mark.gpu
def test_c3_with_auxiliary_head_182_hidden_channels_matches_oracle():
    """DeepLabv3+/xception65 with SOLVER.AUX True (VERDICT r05 Missing #4): the reference builds
    _FCNHead(728, nclass) with 728 // 4 = 182 hidden channels
    (/root/reference/segmentron/models/deeplabv3_plus.py:29-30, modules/module.py:13-26), not a
    multiple of the 16-byte channel vector; the HIP module pads them to 184 inside and keeps the
    reference's state_dict (tests/test_host_api.py checks the schema).  One fp32 train step at
    65x129 against the CPU oracle with the auxiliary loss (weight 0.4): 1e-3 bars."""
    import segmentron_amd
    from segmentron_amd.config import cfg, reset_cfg
    from oracle import parity as OP
    reset_cfg()
    cfg.update_from_list(C3_OVERRIDES + ["SOLVER.AUX", "True"])
    cfg.PHASE = "test"
    cfg.check_and_freeze()
    segmentron_amd.set_compute_dtype(torch.float32)
    model = segmentron_amd.get_segmentation_model()
    assert model.auxlayer.block[0].weight.shape[0] == 184
    sd = synth.synth_like(model.state_dict(), seed=3, conditioned=True)
    assert sd["auxlayer.block.0.weight"].shape[0] == 182
    del model
    x, y = OP.inputs(2, 65, 129, seed=3)
    ref = OP.oracle_step(sd, x, y, torch.float64, aux=True)
    got = OP.hip_step("fp32", sd, x, y)
    c = OP.compare(got, ref)
    aux_keys = [k for k in ref["grads"] if k.startswith("auxlayer.")]
    assert len(aux_keys) == 5 and all(k in got["grads"] for k in aux_keys)
    pad = got["grads"]["auxlayer.block.0.weight"][182:].abs().max().item()
    print("PARITY-COND c3 + aux head (182 -> 184 hidden channels) fp32 65x129: loss rel %.2e logits "
          "max-rel %.2e gradients global rel %.2e; padded rows' gradient %.1e"
          % (c["loss_rel"], c["logits_maxrel"], c["grad_global_rel"], pad))
    assert c["finite"] and c["grad_tensors_missing"] == 0
    assert c["loss_rel"] < 1e-3 and c["logits_maxrel"] < 1e-3 and c["grad_global_rel"] <= 1e-3
    assert pad == 0.0  # the padded channels carry no gradient
    b16 = OP.compare(OP.hip_step("bf16", sd, x, y), ref)  # (the throughput path runs it too)
    m = BF16_MEASURED_LARGE["c3_aux_65x129"]  # (1.3 x measured, see BF16_MEASURED)
    print("PARITY-COND c3 + aux head bf16: logits L2-rel %.3e gradient cosine %.5f"
          % (b16["logits_l2rel"], b16["grad_cosine"]))
    assert b16["finite"] and b16["grad_tensors_missing"] == 0
    assert b16["logits_l2rel"] <= 1.3 * m["logits_l2"], (b16["logits_l2rel"], m)
    assert 1.0 - b16["grad_cosine"] <= 1.3 * (1.0 - m["cos"]) + 2e-5, (b16["grad_cosine"], m)
    reset_cfg()


@pytest.mark.gpu
def test_ccnet_resnet50_with_group_norm_in_encoder_and_heads_matches_oracle():
    """CCNet passes its norm layer on to the heads (/root/reference/segmentron/models/ccnet.py:
    21-23,45,61-71): under BN_TYPE 'GN' the criss-cross head, the bottleneck behind the concat and
    the auxiliary _FCNHead are GroupNorm too — gradients reach the GroupNorm kernels as channel
    slices of concat buffers.  One fp32 train step at 65 x 97 against the float64 oracle."""
    import segmentron_amd
    from segmentron_amd.config import cfg, reset_cfg
    from oracle import parity as OP
    reset_cfg()
    cfg.update_from_list(["DATASET.NAME", "cityscape", "MODEL.MODEL_NAME", "CCNet",
                          "MODEL.BACKBONE", "resnet50", "MODEL.OUTPUT_STRIDE", "16",
                          "MODEL.BN_TYPE", "GN", "SOLVER.AUX", "True", "SOLVER.AUX_WEIGHT",
                          str(AUX_WEIGHT), "TRAIN.BACKBONE_PRETRAINED", "False"])
    cfg.PHASE = "test"
    cfg.check_and_freeze()
    segmentron_amd.set_compute_dtype(torch.float32)
    model = segmentron_amd.get_segmentation_model()
    gns = [n for n, m in model.named_modules() if isinstance(m, torch.nn.GroupNorm)]
    assert not any(isinstance(m, torch.nn.BatchNorm2d) for m in model.modules())
    assert any(n.startswith("head.") for n in gns) and any(n.startswith("auxlayer.") for n in gns)
    state = model.state_dict()
    names = set(state) | set(n + ".running_mean" for n in gns)
    sd = {k: synth.synth_tensor(k, tuple(v.shape), seed=7, conditioned=True, all_keys=names)
          for k, v in state.items()}
    del model
    x, y = OP.inputs(2, 65, 97, seed=7)
    ref = OP.oracle_step(sd, x, y, torch.float64, oracle_fn="ccnet_resnet", output_stride=16,
                         aux=True, eps_encoder=None, norm="GN")
    f32 = OP.compare(OP.hip_step("fp32", sd, x, y, eps_encoder=None), ref)
    print("PARITY-COND CCNet/resnet50 GroupNorm (encoder + heads) 65x97: fp32 loss rel %.2e logits "
          "max-rel %.2e gradients global rel %.2e"
          % (f32["loss_rel"], f32["logits_maxrel"], f32["grad_global_rel"]))
    assert f32["finite"] and f32["grad_tensors_missing"] == 0
    assert f32["loss_rel"] < 1e-3 and f32["logits_maxrel"] < 1e-3 and f32["grad_global_rel"] <= 1e-3
    _assert_fp32_argmax(f32, "CCNet GroupNorm")
    reset_cfg()


@pytest.mark.gpu
def test_pspnet_resnet50_with_group_norm_matches_oracle():
    """cfg.MODEL.BN_TYPE 'GN' (VERDICT r05 Missing #3): the reference maps it to
    nn.GroupNorm(min(32, C), C) in every norm_layer slot (/root/reference/segmentron/modules/
    batch_norm.py:105-108,129; models/segbase.py:23) — PSPNet / resnet50 (every width divisible
    by 32; xception's 728 channels are not: torch itself refuses GroupNorm(32, 728)), output
    stride 8, auxiliary head, the pyramid bins' 1 x 1 .. 6 x 6 maps.  One train step at 65 x 97
    against the float64 oracle (F.group_norm): fp32 kernels 1e-3 on loss, logits and the global
    gradient; bf16: finite, logits L2-rel / gradient cosine recorded bars."""
    import segmentron_amd
    from segmentron_amd.config import cfg, reset_cfg
    from oracle import parity as OP
    reset_cfg()
    cfg.update_from_list(["DATASET.NAME", "cityscape", "MODEL.MODEL_NAME", "PSPNet",
                          "MODEL.BACKBONE", "resnet50", "MODEL.OUTPUT_STRIDE", "8",
                          "MODEL.BN_TYPE", "GN", "SOLVER.AUX", "True", "SOLVER.AUX_WEIGHT",
                          str(AUX_WEIGHT), "TRAIN.BACKBONE_PRETRAINED", "False"])
    cfg.PHASE = "test"
    cfg.check_and_freeze()
    segmentron_amd.set_compute_dtype(torch.float32)
    model = segmentron_amd.get_segmentation_model()
    gns = [n for n, m in model.named_modules() if isinstance(m, torch.nn.GroupNorm)]
    # (as in the reference, whose PSPNet builds _PSPHead / _FCNHead without norm_layer —
    # pspnet.py:23-25 — the heads keep BatchNorm2d: 53 GroupNorms in the encoder, 6 BatchNorms)
    bns = [n for n, m in model.named_modules() if isinstance(m, torch.nn.BatchNorm2d)]
    assert len(gns) == 53 and all(n.startswith("encoder.") for n in gns)
    assert len(bns) == 6 and not any(n.startswith("encoder.") for n in bns)
    state = model.state_dict()
    # the conditioned state's rules key on BatchNorm buffers (oracle/synth.py): name them for the
    # GroupNorm affines so that they get the same treatment (small gamma on the last norm of a
    # residual branch, beta = +2 gamma elsewhere)
    names = set(state) | set(n + ".running_mean" for n in gns)
    sd = {k: synth.synth_tensor(k, tuple(v.shape), seed=5, conditioned=True, all_keys=names)
          for k, v in state.items()}
    del model
    H, W = 65, 97
    x, y = OP.inputs(2, H, W, seed=5)
    ref = OP.oracle_step(sd, x, y, torch.float64, oracle_fn="pspnet_resnet", output_stride=8,
                         aux=True, eps_encoder=None, norm="GN")
    f32 = OP.compare(OP.hip_step("fp32", sd, x, y, eps_encoder=None), ref)
    b16 = OP.compare(OP.hip_step("bf16", sd, x, y, eps_encoder=None), ref)
    print("PARITY-COND PSPNet/resnet50 GroupNorm %dx%d: fp32 loss rel %.2e logits max-rel %.2e "
          "gradients global rel %.2e | bf16 logits L2-rel %.3e gradient cosine %.5f"
          % (H, W, f32["loss_rel"], f32["logits_maxrel"], f32["grad_global_rel"],
             b16["logits_l2rel"], b16["grad_cosine"]))
    assert f32["finite"] and f32["grad_tensors_missing"] == 0
    assert f32["loss_rel"] < 1e-3 and f32["logits_maxrel"] < 1e-3 and f32["grad_global_rel"] <= 1e-3
    _assert_fp32_argmax(f32, "PSPNet GroupNorm")
    assert b16["finite"] and b16["grad_tensors_missing"] == 0
    m = BF16_MEASURED_LARGE["pspnet_gn_65x97"]  # (1.3 x measured, see BF16_MEASURED)
    assert b16["logits_l2rel"] <= 1.3 * m["logits_l2"], (b16["logits_l2rel"], m)
    assert 1.0 - b16["grad_cosine"] <= 1.3 * (1.0 - m["cos"]) + 2e-5, (b16["grad_cosine"], m)
    reset_cfg()
