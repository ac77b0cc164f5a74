"""FCN / PSPNet on ResNet backbones and DeepLabv3+ on MobileNetV2 (BASELINE configs C1, C4, C2):
CPU  — the oracle reproduces the fixtures generated from the reference; the module tree has the
       reference's state_dict schema;
GPU  — the HIP fp32 path matches the reference fixture (eval logits 1e-3 + argmax, train loss /
       logits 1e-3, gradients as accurate w.r.t. the fp64 oracle as the CPU fp32 path)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import synth, torch_ref

CASES = {
    "c1": dict(model="FCN", backbone="resnet101", os=16, aux=False, fn="fcn_resnet", hw=(65, 97),
               aux_weight=0.4),
    "c4": dict(model="PSPNet", backbone="resnet101", os=8, aux=True, fn="pspnet_resnet", hw=(49, 65),
               aux_weight=0.4),
    "c2": dict(model="DeepLabV3_Plus", backbone="mobilenet_v2", os=16, aux=False,
               fn="deeplab_mobilenet", hw=(65, 97), aux_weight=0.4, tie_delta=1e-5,
               over=["MODEL.DEEPLABV3_PLUS.USE_ASPP", "False",
                     "MODEL.DEEPLABV3_PLUS.ENABLE_DECODER", "False"]),
    "c5": dict(model="HRNet", backbone="hrnet_w18_small_v1", os=16, aux=False, fn="hrnet_seg",
               hw=(64, 128), aux_weight=0.4, momentum=0.01, tie_delta=1e-5,
               yaml="configs/cityscapes_hrnet_w18_small_v1.yaml"),
    # CCNet: fixtures from the reference's own module tree with the CUDA extension
    # stood in by the oracle's restatement of ca_cuda.cu (oracle/gen_golden_more.py c6)
    "c6": dict(model="CCNet", backbone="resnet101", os=16, aux=False, fn="ccnet_resnet",
               hw=(65, 97), aux_weight=0.4),
    # DANet (SURVEY §8 f4 tail; configs/cityscapes_danet_resnet.yaml: OS8, multi-grid layer4):
    # three outputs (fused / position / channel heads), all weighted into the loss
    "c8": dict(model="DANet", backbone="resnet101", os=8, aux=False, fn="danet_resnet",
               hw=(49, 65), aux_weight=0.4, nout=3, multi_dilation=[4, 8, 16],
               over=["MODEL.DANET.MULTI_GRID", "True", "MODEL.DANET.MULTI_DILATION", "[4, 8, 16]"]),
    # cfg.MODEL.BN_TYPE 'GN' (r06; modules/batch_norm.py:105-108,129): GroupNorm(min(32, C), C) in
    # the encoder, BatchNorm2d in the heads (the reference builds them without norm_layer,
    # pspnet.py:23-25); fixtures from the reference itself (oracle/gen_golden_more.py c9)
    "c9": dict(model="PSPNet", backbone="resnet50", os=8, aux=True, fn="pspnet_resnet",
               hw=(49, 65), aux_weight=0.4, over=["MODEL.BN_TYPE", "GN"], norm="GN",
               tie_delta=1e-5),
    # Fast-SCNN (SURVEY §8 f4 tail; configs/cityscapes_fast_scnn.yaml: AUX True, BN momentum 0.01)
    "c7": dict(model="FastSCNN", backbone="", os=16, aux=True, fn="fast_scnn", hw=(192, 192),
               aux_weight=0.4, momentum=0.01, tie_delta=1e-5),
}
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _state(tag):
    keys = json.load(open(os.path.join(GOLDEN, tag + "_state_keys.json")))["keys"]
    sd = synth.synth_state_dict([(k, tuple(s)) for k, s in keys], seed=0)
    calib = np.load(os.path.join(GOLDEN, tag + "_bn_calib.npz"))
    for k in calib.files:
        sd[k] = torch.from_numpy(calib[k])
    for k in sd:
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.long)
    return sd


def _cfg(tag):
    from segmentron_amd.config import cfg, reset_cfg
    c = CASES[tag]
    reset_cfg()
    if "yaml" in c:
        cfg.update_from_file(os.path.join(ROOT, c["yaml"]))
    cfg.update_from_list(["DATASET.NAME", "cityscape", "MODEL.MODEL_NAME", c["model"]]
                         + (["MODEL.BACKBONE", c["backbone"]] if c["backbone"] else []) +
                         ["MODEL.OUTPUT_STRIDE", str(c["os"]),
                          "SOLVER.AUX", str(c["aux"]), "SOLVER.AUX_WEIGHT", str(c["aux_weight"]),
                          "TRAIN.BACKBONE_PRETRAINED", "False"] + c.get("over", []))
    cfg.PHASE = "test"
    cfg.check_and_freeze()
    return cfg


class _shifted_relu:
    """Run the oracle with ReLU(x) = x * (x > shift): `shift` = +-delta flips every ReLU whose
    pre-activation lies within delta of zero — the oracle's own sensitivity to near-ties."""

    def __init__(self, shift):
        self.shift = shift

    def __enter__(self):
        import torch.nn.functional as TF
        self.TF, self.orig, self.orig6 = TF, TF.relu, TF.relu6
        if self.shift:
            sh = self.shift
            TF.relu = lambda t, inplace=False: t * (t > sh)
            # ReLU6: both knees move (lower at 0, upper at 6)
            TF.relu6 = lambda t, inplace=False: torch.where(t < 6 + sh, t * (t > sh),
                                                            torch.full_like(t, 6.0))

    def __exit__(self, *a):
        self.TF.relu, self.TF.relu6 = self.orig, self.orig6


def _sub(t, fixture):
    """Fixtures of large outputs store every `sub`-th pixel (oracle/gen_golden_more.py)."""
    k = int(fixture["sub"]) if "sub" in fixture.files else 1
    return t[..., ::k, ::k]


def _oracle(tag, sd, x, training, dtype=torch.float32, y=None, relu_shift=0.0):
    with _shifted_relu(relu_shift):
        return _oracle_impl(tag, sd, x, training, dtype, y)


def _oracle_impl(tag, sd, x, training, dtype=torch.float32, y=None):
    c = CASES[tag]
    s = {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    s = torch_ref.clone_state(s, requires_grad=training)
    net = torch_ref.OracleNet(s, training=training, output_stride=c["os"], aux=c["aux"], drop_p=0.0,
                              momentum=c.get("momentum"), multi_dilation=c.get("multi_dilation"),
                              norm=c.get("norm", "BN"))
    outs = getattr(net, c["fn"])(x.to(dtype))
    if not training:
        return outs, None, None
    loss = torch_ref.mix_softmax_ce(outs, y, aux_weight=c["aux_weight"])
    loss.backward()
    _oracle.last_state = s
    return outs, loss.item(), {k: v.grad for k, v in s.items() if v.grad is not None}


@pytest.mark.parametrize("tag", list(CASES))
def test_oracle_reproduces_reference_fixture(tag):
    sd = _state(tag)
    H, W = CASES[tag]["hw"]
    x = synth.synth_images(2, H, W, seed=0)
    y = synth.synth_targets(2, H, W, seed=0)
    with torch.no_grad():
        outs, _, _ = _oracle(tag, sd, x, False)
    g = np.load(os.path.join(GOLDEN, tag + "_eval.npz"))
    assert torch.allclose(_sub(outs[0], g), torch.from_numpy(g["logits"]), rtol=1e-4, atol=1e-4)
    outs, loss, grads = _oracle(tag, sd, x, True, y=y)
    t = np.load(os.path.join(GOLDEN, tag + "_train.npz"))
    assert abs(loss - float(t["loss"])) < 1e-5
    for k, n in zip([str(k) for k in t["grad_norm_keys"]], t["grad_norms"]):
        assert abs(float(grads[k].double().norm()) - n) <= 1e-3 * max(n, 1e-6) + 1e-9, k


@pytest.mark.parametrize("tag", list(CASES))
def test_state_dict_schema(tag):
    import segmentron_amd
    from segmentron_amd.config import reset_cfg
    _cfg(tag)
    model = segmentron_amd.get_segmentation_model()
    ref = json.load(open(os.path.join(GOLDEN, tag + "_state_keys.json")))
    assert [(k, list(v.shape)) for k, v in model.state_dict().items()] == \
        [(k, list(s)) for k, s in ref["keys"]]
    assert sum(p.numel() for p in model.parameters()) == ref["n_params"]
    reset_cfg()


def test_fcn_resnet18_fails_like_the_reference():
    """the FCN head hard-codes 2048 input channels (fcn.py:16)."""
    import segmentron_amd
    from segmentron_amd.config import cfg, reset_cfg
    reset_cfg()
    cfg.update_from_list(["DATASET.NAME", "cityscape", "MODEL.MODEL_NAME", "FCN", "MODEL.BACKBONE",
                          "resnet18", "TRAIN.BACKBONE_PRETRAINED", "False"])
    cfg.PHASE = "test"
    cfg.check_and_freeze()
    model = segmentron_amd.get_segmentation_model()
    assert model.head.block[0].in_channels == 2048 and model.encoder.last_inp_channels == 512
    reset_cfg()


def _rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max()).item()


def _build_hip(tag, dtype, train):
    import segmentron_amd
    _cfg(tag)
    segmentron_amd.set_compute_dtype(dtype)
    model = segmentron_amd.get_segmentation_model()
    sd = _state(tag)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().train(train)
    for m in model.modules():
        if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)):
            m.p = 0.0
        # what segmentron/solver/optimizer.py:8-30 does after construction
        if isinstance(m, torch.nn.BatchNorm2d) and CASES[tag].get("momentum") is not None:
            m.momentum = CASES[tag]["momentum"]
    return model, sd


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(CASES))
def test_hip_eval_fp32_matches_reference_fixture(tag):
    model, _ = _build_hip(tag, torch.float32, False)
    H, W = CASES[tag]["hw"]
    x = synth.synth_images(2, H, W, seed=0)
    with torch.no_grad():
        outs = model(x.cuda())
    assert len(outs) == CASES[tag].get("nout", (3 if tag == "c7" else 2) if CASES[tag]["aux"] else 1)
    g = np.load(os.path.join(GOLDEN, tag + "_eval.npz"))
    logits = _sub(outs[0].cpu(), g)
    rel = _rel(logits, torch.from_numpy(g["logits"]))
    print("%s eval fp32 max-rel vs reference fixture: %.3e" % (tag, rel))
    ref = torch.from_numpy(g["logits"])
    bad = logits.argmax(1) != ref.argmax(1)
    top2 = ref.topk(2, dim=1).values
    gap = (top2[:, 0] - top2[:, 1])[bad]
    print("%s argmax mismatches: %d of %d (largest reference top-2 gap among them %.2e)"
          % (tag, int(bad.sum()), bad.numel(), gap.max().item() if gap.numel() else 0.0))
    assert rel < 1e-3
    # masks identical except at genuine ties of the reference itself (top-2 gap below the 1e-3 bar)
    assert gap.numel() == 0 or gap.max().item() < 1e-3 * ref.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(CASES))
def test_hip_train_fp32_matches_reference(tag):
    c = CASES[tag]
    model, sd = _build_hip(tag, torch.float32, True)
    H, W = c["hw"]
    x = synth.synth_images(2, H, W, seed=0)
    y = synth.synth_targets(2, H, W, seed=0)
    outs = model(x.cuda())
    loss = torch.nn.functional.cross_entropy(outs[0], y.cuda(), ignore_index=-1)
    for o in outs[1:]:
        loss = loss + c["aux_weight"] * torch.nn.functional.cross_entropy(o, y.cuda(), ignore_index=-1)
    loss.backward()
    t = np.load(os.path.join(GOLDEN, tag + "_train.npz"))
    rel = _rel(_sub(outs[0].detach().cpu(), t), torch.from_numpy(t["logits"]))
    print("%s train fp32: loss %.6f (fixture %.6f) logits max-rel %.3e" % (tag, loss.item(), float(t["loss"]), rel))
    assert abs(loss.item() - float(t["loss"])) < 1e-3 * float(t["loss"]) and rel < 1e-3
    _, _, g64 = _oracle(tag, sd, x, True, torch.float64, y)
    _, _, g32 = _oracle(tag, sd, x, True, torch.float32, y)
    # running statistics after the step: momentum, unbiased variance, conv-bias offset
    ostate, msd = _oracle.last_state, model.state_dict()
    for k, v in msd.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            ref = ostate[k].detach()
            tol = 1e-3 * ref.abs().max().item() + 1e-6
            assert (v.cpu() - ref).abs().max().item() <= tol, k
    for k in t.files:
        if k.startswith("stat::"):
            ref = torch.from_numpy(t[k])
            assert (msd[k[6:]].cpu() - ref).abs().max().item() <= 1e-3 * ref.abs().max().item() + 1e-6, k
    # torch nn.BatchNorm2d: every training forward advances num_batches_tracked of every BatchNorm
    # it ran (CCNet's never did) — EVERY counter against the oracle's own
    counters = [k for k in msd if k.endswith("num_batches_tracked")]
    assert counters and any(int(ostate[k]) == 1 for k in counters)
    bad = [(k, int(msd[k]), int(ostate[k])) for k in counters if int(msd[k]) != int(ostate[k])]
    assert not bad, bad[:5]
    from segmentron_amd import functional as HF
    assert not HF._PENDING_COUNTERS and HF._COUNTER_SCOPE[0] == 0
    params = dict(model.named_parameters())
    # Near-tie ReLUs: a pre-activation within fp32 rounding of zero may land on the other side in
    # a different (equally valid) fp32 evaluation order; one flip at HRNet's 4x8 / 2x4 branches
    # moves every upstream gradient by percents (c5, seed 0: stage4 row 2, channel 46 has
    # pre-activation -7.6e-7 in fp64, -6.9e-6 in the reference's own fp32 run).  The oracle's
    # sensitivity to flipping ITS near-ties (|x| < delta) bounds what such flips may cost.
    sens = {k: 0.0 for k in g64}
    if c.get("tie_delta"):
        for sh in (c["tie_delta"], -c["tie_delta"]):
            _, _, gs = _oracle(tag, sd, x, True, torch.float64, y, relu_shift=sh)
            for k in g64:
                sens[k] += (gs[k] - g64[k]).norm().item()
    nh = nc = den = ns = 0.0
    worst = (0.0, "")
    allw = []
    for k, t64 in g64.items():
        assert params[k].grad is not None, k
        gh = params[k].grad.detach().cpu().double()
        assert torch.isfinite(gh).all(), k
        eh, ec, n64 = (gh - t64).norm().item(), (g32[k].double() - t64).norm().item(), t64.norm().item()
        nh, nc, den, ns = nh + eh ** 2, nc + ec ** 2, den + n64 ** 2, ns + sens[k] ** 2
        bound = 4 * ec + 1e-3 * n64 if n64 > 10 * ec else 20 * ec + 1e-12
        bound += sens[k]
        worst = max(worst, (eh / max(bound, 1e-30), k))
        allw.append((eh / max(bound, 1e-30), k, eh, ec, n64))
    for w in sorted(allw, reverse=True)[:8]:
        print("   %-50s ratio %.2f err_hip %.3e err_cpu32 %.3e |g64| %.3e" % (w[1], w[0], w[2], w[3], w[4]))
    if os.environ.get("SEG_DEBUG_GRADS"):
        for w in allw:
            print("   ALL %-50s ratio %.2f err_hip %.3e |g64| %.3e" % (w[1], w[0], w[2], w[4]))
    print("%s gradients vs fp64 oracle: global rel err HIP %.3e, CPU-fp32 %.3e; worst ratio %.2f (%s)"
          % (tag, (nh / den) ** 0.5, (nc / den) ** 0.5, worst[0], worst[1]))
    print("%s near-tie sensitivity of the fp64 oracle (global rel): %.3e" % (tag, (ns / den) ** 0.5))
    assert (nh / den) ** 0.5 <= 3 * (nc / den) ** 0.5 + 1e-4 + (ns / den) ** 0.5
    # Per tensor: as accurate as the CPU fp32 path (4x its distance to fp64 + 1e-3).  PSPNet's
    # pyramid applies training-mode BatchNorm over only N*o*o = 2..72 pooled samples, which
    # amplifies fp32 rounding in the forward (~8e-5 at the head conv output, every kernel checked
    # to 1e-6 in isolation, tools/debug_bn_bwd.py) and flips a few ReLU masks at near-zero
    # pre-activations; one flip moves a channel's gradient sum by ~1/sqrt(63).  Hence: at most
    # 10 % of the tensors may miss the tight bound and none may miss 4x + 3e-2.
    over = [w for w in allw if w[0] > 1.0]
    assert len(over) <= 0.10 * len(allw), (len(over), len(allw))
    for _, k, eh, ec, n64 in over:
        assert eh <= 4 * ec + 3e-2 * n64, (k, eh, ec, n64)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(CASES))
def test_hip_bf16_runs_and_is_finite(tag):
    model, _ = _build_hip(tag, torch.bfloat16, True)
    H, W = CASES[tag]["hw"]
    x = synth.synth_images(2, H, W, seed=0)
    y = synth.synth_targets(2, H, W, seed=0)
    outs = model(x.cuda())
    loss = sum(torch.nn.functional.cross_entropy(o, y.cuda(), ignore_index=-1) for o in outs)
    loss.backward()
    assert torch.isfinite(loss)
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)


@pytest.mark.gpu
@pytest.mark.parametrize("nb", [2, 3, 4])
def test_hrnet_module_and_head_gradients_tight(nb):
    """One HighResolutionModule (+ transition-style deferred input + the head's bilinear
    align_corners=False concat) on random inputs vs the fp64 oracle: forward, input gradients and
    every parameter gradient to 1e-4 — the whole-model check above can only be as tight as the
    net's ReLU near-ties allow, this one has none."""
    import copy

    import torch.nn as nn
    import torch.nn.functional as TF

    import segmentron_amd
    from segmentron_amd import functional as F
    from segmentron_amd.models.backbones import hrnet as HR
    segmentron_amd.set_compute_dtype(torch.float32)
    ch = [16, 32, 64, 128][:nb]
    mod = HR.HighResolutionModule(nb, HR.BasicBlock, [2] * nb, list(ch), list(ch), "SUM", True)
    sd = synth.synth_like(mod.state_dict(), seed=3)
    mod.load_state_dict(sd)
    mod = mod.cuda().train()
    gen = torch.Generator().manual_seed(11)
    N, H, W = 2, 16, 32
    xs = [torch.randn(N, c, H >> i, W >> i, generator=gen) for i, c in enumerate(ch)]
    gcat = torch.randn(N, sum(ch), H, W, generator=gen)
    tconv = nn.Conv2d(ch[-2], ch[-1], 3, 2, 1, bias=False)
    tbn = nn.BatchNorm2d(ch[-1])
    tconv_o, tbn_o = copy.deepcopy(tconv).double(), copy.deepcopy(tbn).double()
    tconv, tbn = tconv.cuda(), tbn.cuda()
    # fp64 oracle
    osd = torch_ref.clone_state({("m." + k): (v.double() if v.is_floating_point() else v)
                                 for k, v in sd.items()}, requires_grad=True)
    net = torch_ref.OracleNet(osd, training=True)
    xo = [t.double().requires_grad_() for t in xs]
    xin = list(xo)
    xin[-1] = torch.relu(tbn_o(tconv_o(xo[-2])))  # like transitionN: conv s2 + BN + ReLU
    yo = torch_ref._hr_module(net, xin, "m")
    cat_o = torch.cat([yo[0]] + [TF.interpolate(t, size=(H, W), mode="bilinear", align_corners=False)
                                 for t in yo[1:]], 1)
    (cat_o * gcat.double()).sum().backward()
    # HIP
    xh = [t.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_() for t in xs]
    ain = [F.Act(t) for t in xh]
    a = F.conv_bn(ain[-2], tconv, tbn)
    a.relu = True
    ain[-1] = a
    yh = mod(ain)
    buf = torch.empty((N, H, W, sum(ch)), device="cuda")
    parts = [F.materialize(yh[0], out=buf[..., :ch[0]], force=True)]
    o = ch[0]
    for a, c in zip(yh[1:], ch[1:]):
        parts.append(F.bilinear(a, (H, W), align_corners=False, out=buf[..., o:o + c]))
        o += c
    cat_h = F.concat_alias(buf, parts)

    def rel(a, b):
        return ((a.double() - b).norm() / b.norm()).item()

    assert rel(cat_h.detach().cpu().permute(0, 3, 1, 2), cat_o.detach()) < 1e-5
    (cat_h * gcat.permute(0, 2, 3, 1).contiguous().cuda()).sum().backward()
    for i in range(nb - 1):
        assert rel(xh[i].grad.cpu().permute(0, 3, 1, 2), xo[i].grad) < 1e-4, i
    for k, p in mod.named_parameters():
        assert rel(p.grad.cpu(), osd["m." + k].grad) < 1e-4, k
    assert rel(tconv.weight.grad.cpu(), tconv_o.weight.grad) < 1e-4
    assert rel(tbn.weight.grad.cpu(), tbn_o.weight.grad) < 1e-4


# ------------------------------------------------------------- BASELINE sizes (configs[1,3,4])
# (batch, H, W) of BASELINE.json's C2 / C4 / C5.  At these sizes every fast-path gate is open
# (M >= 4096 / >= 65536 pixels): the direct-to-LDS GEMMs incl. their KxK form on the PSP head
# (C = 4096) and ResNet layer3/4, the 256x64 tile, the persistent multi-tile depthwise kernels,
# the wide bilinear kernels — none of which the 49x65 .. 65x97 fixtures above select.
FULL = {"c2": (1, 1024, 2048), "c4": (1, 1025, 2049), "c5": (2, 1024, 2048)}


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(FULL))
def test_hip_eval_fp32_baseline_size_matches_oracle(tag):
    """fp32 HIP eval vs the CPU oracle at the BASELINE geometry: logits within 1e-3 relative,
    argmax masks identical up to the oracle's own near-ties (VERDICT r02 next-#1a)."""
    from _util import tie_tolerant_argmax_check
    B, H, W = FULL[tag]
    model, sd = _build_hip(tag, torch.float32, False)
    x = synth.synth_images(B, H, W, seed=21)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    with torch.no_grad():
        got = model(x.cuda())[0].cpu()
        ref = _oracle(tag, sd, x, False)[0][0]
    assert tuple(got.shape) == tuple(ref.shape) == (B, 19, H, W)
    rel = _rel(got, ref)
    n_tie = tie_tolerant_argmax_check(got, ref, "%s eval %dx%d" % (tag, H, W))
    print("PARITY %s eval fp32 %dx%dx%d max-rel vs oracle %.3e; argmax: %d near-tie pixels of %d "
          "differ" % (tag, B, H, W, rel, n_tie, B * H * W))
    assert rel < 1e-3


@pytest.mark.gpu
def test_hip_hrnet_batch16_images_equal_their_batch2_run():
    """C5 is quoted at batch 16: eval-mode outputs of an image must not depend on its batch
    neighbours nor on the tile schedule a larger batch selects — bit-for-bit, fp32 and bf16."""
    H, W = 1024, 2048
    x = synth.synth_images(16, H, W, seed=22)
    for dtype in (torch.float32, torch.bfloat16):
        model, _ = _build_hip("c5", dtype, False)
        with torch.no_grad():
            full = model(x.cuda())[0]
            assert tuple(full.shape) == (16, 19, H, W) and torch.isfinite(full).all()
            for lo in (0, 6, 14):
                part = model(x[lo:lo + 2].cuda())[0]
                assert torch.equal(full[lo:lo + 2], part), (str(dtype), lo)
                del part
        del full, model
        torch.cuda.empty_cache()
