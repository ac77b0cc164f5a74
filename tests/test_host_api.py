"""CPU: host-side mirror of the reference interface — cfg semantics, registries, module tree /
state_dict schema, and the C-ABI library surface."""
import ctypes
import json
import os
import re

import pytest
import torch


def test_state_dict_schema_equals_reference(c3_cfg, golden_dir):
    import segmentron_amd
    model = segmentron_amd.get_segmentation_model()
    ref = json.load(open(os.path.join(golden_dir, "c3_state_keys.json")))
    got = [(k, list(v.shape)) for k, v in model.state_dict().items()]
    assert got == [(k, list(s)) for k, s in ref["keys"]]
    assert sum(p.numel() for p in model.parameters()) == ref["n_params"] == 41054899
    assert model.decoder == ["head"] and model.nclass == 19 and model.aux is False
    assert model.backbone == "xception65"


def test_bn_children_stay_batchnorm_and_convert_to_syncbn(c3_cfg):
    import segmentron_amd
    import torch.nn as nn
    model = segmentron_amd.get_segmentation_model()
    bns = [m for m in model.modules() if isinstance(m, nn.BatchNorm2d)]
    assert len(bns) == 146 and sum(b.num_features for b in bns) == 101400  # SURVEY F11 census
    # what solver/optimizer.py:8-11 does after construction must reach the kernels (F6)
    for _, m in model.encoder.named_modules():
        if isinstance(m, nn.BatchNorm2d):
            m.eps = 1e-3
    assert model.encoder.block4.sep_conv1.block.bn_depth.eps == 1e-3
    sync = nn.SyncBatchNorm.convert_sync_batchnorm(model)
    assert isinstance(sync.encoder.block4.sep_conv1.block.bn_depth, nn.SyncBatchNorm)
    assert list(sync.state_dict().keys()) == list(model.state_dict().keys())


def test_cfg_semantics():
    from segmentron_amd.config import cfg, reset_cfg
    reset_cfg()
    cfg.update_from_list(["TEST.CROP_SIZE", "(1025, 2049)", "MODEL.MODEL_NAME", "DeepLabV3_Plus",
                          "MODEL.BACKBONE", "xception65", "DATASET.NAME", "cityscape"])
    assert cfg.TEST.CROP_SIZE == (1025, 2049)  # strings are literal_eval'ed
    with pytest.raises(KeyError):
        cfg.update_from_list(["MODEL.NO_SUCH_KEY", "1"])
    with pytest.raises(ValueError):
        cfg.update_from_list(["MODEL.BACKBONE"])
    cfg.check_and_freeze()
    assert "DANET" not in cfg.MODEL and "DEEPLABV3_PLUS" in cfg.MODEL and cfg.TIME_STAMP
    with pytest.raises(AttributeError):
        cfg.SEED = 3
    reset_cfg()
    assert cfg.SEED == 1024 and not cfg.is_immutable()


def test_cfg_reads_reference_yaml_if_present():
    path = "/root/reference/configs/cityscapes_deeplabv3_plus.yaml"
    if not os.path.exists(path):
        pytest.skip("reference tree not present")
    from segmentron_amd.config import cfg, reset_cfg
    reset_cfg()
    cfg.update_from_file(path)
    assert cfg.MODEL.BACKBONE == "xception65" and cfg.MODEL.BN_EPS_FOR_ENCODER == 1e-3
    assert cfg.TEST.CROP_SIZE == (1025, 2049) and cfg.SOLVER.LR == 0.02
    reset_cfg()


def test_registries():
    from segmentron_amd.models.model_zoo import MODEL_REGISTRY
    from segmentron_amd.models.backbones import BACKBONE_REGISTRY
    assert "DeepLabV3_Plus" in MODEL_REGISTRY.get_list()
    assert "xception65" in BACKBONE_REGISTRY.get_list()
    with pytest.raises(KeyError):
        MODEL_REGISTRY.get("deeplabv3_plus")  # lookup is case-sensitive (model_zoo.py:22)
    with pytest.raises(AssertionError):
        MODEL_REGISTRY.register(name="DeepLabV3_Plus")(object)


def test_segmentron_alias_package(c3_cfg):
    from segmentron.config import cfg as c2
    from segmentron.models.model_zoo import get_segmentation_model, MODEL_REGISTRY  # noqa: F401
    from segmentron.models.backbones import get_segmentation_backbone  # noqa: F401
    from segmentron.modules import SeparableConv2d, _ASPP, _ConvBNReLU, get_norm  # noqa: F401
    assert c2 is c3_cfg


def test_abi_library_exports_every_declared_symbol():
    from segmentron_amd import _lib
    protos = _lib.parse_header()
    assert len(protos) >= 20
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libsegmentron_hip.so not built")
    dll = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(dll, name), name
    # and nothing exported that the header does not declare
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True,
                         text=True).stdout
    exported = set(re.findall(r" T (seg_\w+)", out))
    assert exported == set(protos), exported ^ set(protos)
    _lib.LIB.load()
    assert _lib.LIB.query("seg_version") >= 1
    assert _lib.LIB.query("seg_conv_gemm_tiles_m", 2, 65, 129) == (2 * 65 * 129 + 127) // 128


def test_product_fails_loudly_without_a_device(c3_cfg):
    import segmentron_amd
    if torch.cuda.is_available():
        pytest.skip("device present")
    model = segmentron_amd.get_segmentation_model().eval()
    with pytest.raises(RuntimeError):
        model(torch.zeros(1, 3, 33, 33))


def test_evaluate_glue_matches_the_reference_fixture():
    """SegBaseModel.evaluate (segbase.py:44-79): multi-scale / flip / pad / crop glue, called
    unbound on a stub forward, vs fixtures generated by the REFERENCE's own method
    (oracle/gen_golden_eval.py) — including the reference's F.pad argument order."""
    import os

    import numpy as np
    import torch

    from conftest import GOLDEN
    from oracle.gen_golden_eval import CASES, Stub, image
    from segmentron_amd.config import cfg, reset_cfg
    from segmentron_amd.models.segbase import SegBaseModel
    gold = np.load(os.path.join(GOLDEN, "evaluate_cases.npz"))
    for name, (h, w), scales, flip, crop in CASES:
        reset_cfg()
        cfg.TEST.SCALES, cfg.TEST.FLIP, cfg.TEST.CROP_SIZE = scales, flip, crop
        with torch.no_grad():
            got = SegBaseModel.evaluate(Stub(), image(h, w))
        ref = torch.from_numpy(gold[name])
        assert got.shape == ref.shape, name
        assert (got - ref).abs().max().item() <= 1e-5 * ref.abs().max().item(), name
    reset_cfg()


def test_cfg_default_tree_equals_the_reference():
    """Every key and default of segmentron/config/settings.py (dumped from the reference by
    oracle/gen_golden_cfg.py) — the contract with the reference's yaml files / command lines."""
    import json
    import os

    from conftest import GOLDEN
    from oracle.gen_golden_cfg import plain
    from segmentron_amd.config import cfg, reset_cfg
    reset_cfg()
    ref = json.load(open(os.path.join(GOLDEN, "cfg_defaults.json")))
    mine = json.loads(json.dumps(plain(dict(cfg)), sort_keys=True))
    assert mine == ref


def test_logits_view_dispatch_and_ddp_visibility():
    """functional.LogitsView (what a training-mode forward returns per head): metadata without
    materialisation, F.cross_entropy routed to the fused kernel only for the supported variant,
    everything else through the materialised tensor, and the low-resolution tensor visible to
    DistributedDataParallel's output scan (tools/train.py:110 find_unused_parameters=True)."""
    import dataclasses
    import torch.nn.functional as TF
    from torch.nn.parallel.distributed import _find_tensors
    from segmentron_amd import functional as F
    lo = torch.randn(2, 5, 7, 19, requires_grad=True)
    v = F.LogitsView(lo, (17, 25), True)
    assert tuple(v.shape) == (2, 19, 17, 25) and v.size(1) == 19 and v.dim() == 4
    assert v.dtype == torch.float32 and v.requires_grad and v._full is None
    assert dataclasses.is_dataclass(v) and [t is lo for t in _find_tensors((v,))] == [True]
    calls = []
    orig_ce, orig_full = F.fused_cross_entropy, F._LogitsFn.apply
    try:
        def fake_ce(lo_, target, out_hw, ignore, align):
            calls.append(("fused", ignore))
            up = TF.interpolate(lo_.permute(0, 3, 1, 2), out_hw, mode="bilinear", align_corners=True)
            return TF.cross_entropy(up, target, ignore_index=ignore)

        def fake_full(lo_, out_hw, align):
            calls.append(("full",))
            return TF.interpolate(lo_.permute(0, 3, 1, 2), out_hw, mode="bilinear", align_corners=align)
        F.fused_cross_entropy, F._LogitsFn.apply = fake_ce, fake_full
        t = torch.randint(0, 19, (2, 17, 25))
        l1 = TF.cross_entropy(v, t, ignore_index=-1)
        l2 = torch.nn.CrossEntropyLoss(ignore_index=-1)(v, t)   # what solver/loss.py:16-46 calls
        assert calls == [("fused", -1), ("fused", -1)] and torch.equal(l1, l2)
        l3 = TF.cross_entropy(v, t, ignore_index=-1, reduction="sum")   # not fusable
        assert calls[-1] == ("full",) and l3.item() > l1.item()
        n = len(calls)
        assert torch.argmax(v, 1).shape == (2, 17, 25) and v[..., :4, :5].shape == (2, 19, 4, 5)
        assert v.detach().shape == (2, 19, 17, 25) and len(calls) == n   # materialised ONCE
        l1.backward()
        assert lo.grad is not None
    finally:
        F.fused_cross_entropy, F._LogitsFn.apply = orig_ce, orig_full


def test_torch_custom_ops_are_registered_with_fake_implementations():
    """torch.ops.segmentron_hip.* (segmentron_amd/torch_ops.py): schemas, and shape / dtype
    inference through FakeTensorMode — no device is touched, so this runs without a GPU."""
    import segmentron_amd  # noqa: F401  registers the operators
    from segmentron_amd import torch_ops
    from torch._subclasses.fake_tensor import FakeTensorMode
    ns = torch.ops.segmentron_hip
    for name in torch_ops.OPS:
        assert hasattr(ns, name), name
    assert "relu_in" in str(ns.conv2d.default._schema)
    with FakeTensorMode():
        x = torch.empty(2, 33, 65, 728, dtype=torch.bfloat16, device="cuda")
        w = torch.empty(1024, 728, 1, 1, device="cuda")
        y = ns.conv2d(x, w, None, 1, 0, 1, False)
        assert tuple(y.shape) == (2, 33, 65, 1024) and y.dtype == torch.bfloat16
        w3 = torch.empty(64, 728, 3, 3, device="cuda")
        assert tuple(ns.conv2d(x, w3, None, 2, 1, 1, True).shape) == (2, 17, 33, 64)
        dx, dw, db = ns.conv2d_backward(x, y, w, 1, 0, 1, False, True)
        assert tuple(dx.shape) == tuple(x.shape) and tuple(dw.shape) == tuple(w.shape)
        assert dw.dtype == torch.float32 and tuple(db.shape) == (1024,)
        wd = torch.empty(728, 1, 3, 3, device="cuda")
        assert tuple(ns.depthwise_conv3x3(x, wd, 2, 1, True).shape) == (2, 17, 33, 728)
        assert tuple(ns.depthwise_conv3x3(x, wd, 1, 12, False).shape) == (2, 33, 65, 728)
        assert tuple(ns.interpolate_bilinear(x, 129, 257, True).shape) == (2, 129, 257, 728)
        lo = torch.empty(2, 257, 513, 19, dtype=torch.bfloat16, device="cuda")
        t = torch.empty(2, 1025, 2049, dtype=torch.long, device="cuda")
        out = ns.upsample_cross_entropy(lo, t, 1025, 2049, -1, True)
        assert tuple(out.shape) == (2,) and out.dtype == torch.float32
        q = torch.empty(2, 49, 49, 64, dtype=torch.bfloat16, device="cuda")
        v = torch.empty(2, 49, 49, 512, dtype=torch.bfloat16, device="cuda")
        gam = torch.empty(1, device="cuda")
        o, att, raw = ns.criss_cross_attention(q, q, v, v, gam)
        assert tuple(o.shape) == tuple(v.shape) and tuple(att.shape) == (2, 49, 49, 97)
        assert att.dtype == torch.float32 and raw.dtype == torch.bfloat16
        dq, dk, dv, dg = ns.criss_cross_attention_backward(o, q, q, v, att, raw, gam)
        assert tuple(dq.shape) == tuple(q.shape) and tuple(dv.shape) == tuple(v.shape)
        cnt = ns.segmentation_counts(torch.empty(2, 19, 65, 129, device="cuda"),
                                     torch.empty(2, 65, 129, dtype=torch.long, device="cuda"), 19)
        assert tuple(cnt.shape) == (59,) and cnt.dtype == torch.int64


def _g4_rows(M, O, kxk=False):
    """csrc/conv_gemm_glds.hip glds_rows_per_tile: rounds on 256 CUs x (tile rows + fixed cost);
    r06: + 224-row tiles (the four-wave kernel, 1x1 only) where they make at most 512 tiles."""
    best = None
    for bm in (256, 224, 192):
        tiles = -(-M // bm) * -(-O // 256)
        if bm == 224 and (kxk or tiles > 512):
            continue
        cost = -(-tiles // 256) * (bm + 96)
        if best is None or cost < best[0]:
            best = (cost, bm)
    return -(-M // best[1])


def test_kernel_selection_queries_of_the_c_library():
    """The host-side dispatch rules of libsegmentron_hip.so that size the partial buffers
    (no device needed): which forward / weight-gradient kernel a geometry runs on decides how
    many statistics rows / split partials the caller must allocate."""
    from segmentron_amd import _lib
    q = _lib.LIB.query
    BF16, F32 = 1, 0
    # xception conv2 [2,513,1025,32] -> 64, 3x3 s1 p1: bf16 = direct halo-tile kernel (one row per
    # persistent block, 512), fp32 = implicit GEMM's 256x64 tile (one row per 256 pixels)
    M = 2 * 513 * 1025
    assert q("seg_conv_gemm_stat_rows", BF16, 2, 513, 1025, 32, 64, 3, 3, 1, 1, 1, 0, 0, 3) == 512
    assert q("seg_conv_gemm_stat_rows", F32, 2, 513, 1025, 32, 64, 3, 3, 1, 1, 1, 0, 0, 3) == (M + 255) // 256
    # HRNet's 16 -> 16 basic blocks (r05): the same kernel with four persistent blocks per CU
    assert q("seg_conv_gemm_stat_rows", BF16, 16, 256, 512, 16, 16, 3, 3, 1, 1, 1, 0, 0, 3) == 1024
    assert q("seg_conv_gemm_stat_rows", BF16, 1, 64, 128, 16, 16, 3, 3, 1, 1, 1, 0, 0, 0) == (64 * 128 + 127) // 128  # < 65536 pixels
    # a bias keeps the conv on the implicit GEMM; dilation 2 likewise
    assert q("seg_conv_gemm_stat_rows", BF16, 2, 513, 1025, 32, 64, 3, 3, 1, 1, 1, 0, 1, 0) == (M + 255) // 256
    assert q("seg_conv_gemm_stat_rows", BF16, 2, 513, 1025, 32, 64, 3, 3, 1, 2, 2, 0, 0, 0) == (M + 255) // 256
    # small map: first-generation 128-pixel tiles; 1x1 with O >= 384: the 256-pixel-tile kernels
    assert q("seg_conv_gemm_stat_rows", BF16, 2, 33, 65, 32, 64, 3, 3, 1, 1, 1, 0, 0, 0) == (2 * 33 * 65 + 127) // 128
    # ... bf16 without prologue / bias: the r05 kernel picks 256- or 192-row tiles, whichever
    # fills the 256 CUs in fewer / shorter rounds, r06 adds 224 rows on the four-wave kernel
    # (728 -> 728 @ 16770 pixels: 225 tiles of 224 rows in ONE round instead of 198 of 256;
    # 728 -> 1024: 352 tiles of 192 rows instead of 264 of 256 or 300 of 224, all two rounds;
    # 1536 -> 1536: 450 of 224, two rounds); with a bias or in fp32: the 256-pixel-tile kernel
    assert q("seg_conv_gemm_stat_rows", BF16, 2, 65, 129, 728, 728, 1, 1, 1, 0, 1, 0, 0, 0) == (2 * 65 * 129 + 223) // 224
    assert q("seg_conv_gemm_stat_rows", BF16, 2, 65, 129, 728, 1024, 1, 1, 1, 0, 1, 0, 0, 0) == (2 * 65 * 129 + 191) // 192
    assert q("seg_conv_gemm_stat_rows", BF16, 2, 65, 129, 1536, 1536, 1, 1, 1, 0, 1, 0, 0, 0) == (2 * 65 * 129 + 223) // 224
    assert _g4_rows(2 * 65 * 129, 728) == 75 and _g4_rows(2 * 65 * 129, 1024) == 88
    # many rounds: the 224-row tile is not offered (304 -> 256 @ 263682 pixels: 1178 tiles)
    Md = 2 * 257 * 513
    assert q("seg_conv_gemm_stat_rows", BF16, 2, 257, 513, 304, 256, 1, 1, 1, 0, 1, 0, 0, 0) == _g4_rows(Md, 256) == (Md + 191) // 192
    assert q("seg_conv_gemm_stat_rows", BF16, 2, 65, 129, 728, 728, 1, 1, 1, 0, 1, 0, 1, 0) == (2 * 65 * 129 + 255) // 256
    assert q("seg_conv_gemm_stat_rows", F32, 2, 65, 129, 728, 728, 1, 1, 1, 0, 1, 0, 0, 0) == (2 * 65 * 129 + 255) // 256
    # a handful of pixels in float32 (ASPP image pooling, PSP bins): one row per 8-pixel chunk
    assert q("seg_conv_gemm_stat_rows", F32, 2, 1, 1, 2048, 256, 1, 1, 1, 0, 1, 0, 0, 0) == 1
    assert q("seg_conv_gemm_stat_rows", F32, 2, 6, 6, 2048, 512, 1, 1, 1, 0, 1, 0, 0, 0) == 9
    assert q("seg_conv_gemm_stat_rows", BF16, 2, 6, 6, 2048, 512, 1, 1, 1, 0, 1, 0, 0, 0) == 1
    assert q("seg_conv_gemm_stat_rows", F32, 2, 6, 6, 2048, 512, 1, 1, 1, 0, 1, 0, 0, 3) == 1
    # ResNet layer3 3x3 (256 -> 256, dilation 2) without a prologue: the direct-to-LDS pipeline
    # as an implicit GEMM (256-pixel tiles); with a pending BatchNorm/ReLU: first generation
    Mr = 2 * 129 * 257
    assert q("seg_conv_gemm_stat_rows", BF16, 2, 129, 257, 256, 256, 3, 3, 1, 2, 2, 0, 0, 0) == _g4_rows(Mr, 256, kxk=True)
    assert q("seg_conv_gemm_stat_rows", BF16, 2, 129, 257, 256, 256, 3, 3, 1, 2, 2, 0, 0, 3) == (Mr + 127) // 128
    # weight gradient: conv2 on the direct kernel (one partial per persistent block); plain 1x1
    # 728x728 on the direct-to-LDS kernel (~one block per CU: 36 tiles x 7 splits)
    assert q("seg_conv_gemm_wgrad_splits", BF16, 2, 513, 1025, 32, 64, 3, 3, 1, 1, 1, 3) == 512
    assert q("seg_conv_gemm_wgrad_splits", BF16, 2, 65, 129, 728, 728, 1, 1, 1, 0, 1, 0) == 7
    # depthwise: stride-2 fused backward (one resident set of 768 blocks over the channel blocks),
    # wide dilation = row chains (images x phases x segments), capped the same way
    assert q("seg_dwconv3x3_s2_grid_y", 128, 2, 513, 1025) == 768 // 4
    assert q("seg_dwconv_grid_y", BF16, 2048, 2, 65, 129, 1, 18, 0) == min(2 * 18 * 1, 768 // 64)
    assert q("seg_dwconv_grid_y", BF16, 2048, 2, 65, 129, 1, 6, 1) == min(2 * 6 * 1, 768 // 64)


def test_frozen_batchnorm_module_contract():
    """get_norm('FrozenBN') — segmentron/modules/batch_norm.py:10-104,107-132: four BUFFERS, the
    reference's state_dict keys, its version-3 loading rule, convert_frozen_batchnorm; 'GN'
    (batch_norm.py:105-108,129) builds nn.GroupNorm(min(32, C), C) with the reference's state_dict
    keys (r06: served by csrc/groupnorm.hip)."""
    import pytest
    from segmentron_amd import functional as F
    from segmentron_amd.modules.batch_norm import FrozenBatchNorm2d, get_norm
    assert get_norm("FrozenBN") is FrozenBatchNorm2d
    gn = get_norm("GN")(256)
    assert isinstance(gn, torch.nn.GroupNorm) and gn.num_groups == 32 and gn.eps == 1e-5 and gn.affine
    assert get_norm("GN")(16).num_groups == 16 and F.is_group_norm(gn)
    assert list(gn.state_dict().keys()) == ["weight", "bias"]
    with pytest.raises(AssertionError):
        get_norm("LN")
    m = FrozenBatchNorm2d(6, eps=1e-3)
    assert list(m.state_dict().keys()) == ["weight", "bias", "running_mean", "running_var"]
    assert not list(m.parameters()) and not F.uses_batch_stats(m.train())
    assert torch.allclose(m.running_var, torch.ones(6) - 1e-3)
    # version < 3 checkpoints stored running_var WITHOUT the eps correction
    old = {"weight": torch.full((6,), 2.0), "bias": torch.zeros(6), "running_mean": torch.ones(6),
           "running_var": torch.full((6,), 4.0)}
    sd = torch.nn.Module.state_dict(m) | {k: v.clone() for k, v in old.items()}
    md = {"": {"version": 2}}
    meta = type(m.state_dict())(sd)
    meta._metadata = md
    m.load_state_dict(meta)
    assert torch.allclose(m.running_var, torch.full((6,), 4.0 - 1e-3))
    # conversion of a BatchNorm tree (running_var + eps is what the frozen module stores)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 1), torch.nn.BatchNorm2d(4, eps=1e-2))
    net[1].running_var.fill_(3.0)
    net[1].weight.data.fill_(0.5)
    fz = FrozenBatchNorm2d.convert_frozen_batchnorm(net)
    assert isinstance(fz[1], FrozenBatchNorm2d) and fz[0] is net[0]
    assert torch.allclose(fz[1].running_var, torch.full((4,), 3.01))
    assert torch.allclose(fz[1].weight, torch.full((4,), 0.5))
    # the reference's formula (batch_norm.py:40-45) is what seg_bn_eval_affine computes
    x = torch.randn(2, 6, 3, 3)
    scale = m.weight * (m.running_var + m.eps).rsqrt()
    ref = x * scale.view(1, -1, 1, 1) + (m.bias - m.running_mean * scale).view(1, -1, 1, 1)
    want = torch.nn.functional.batch_norm(x, m.running_mean, m.running_var, m.weight, m.bias,
                                          False, 0.0, m.eps)
    assert torch.allclose(ref, want, atol=1e-6)


def test_weight_cache_bookkeeping_on_the_host():
    """functional.cached_pack / packed-weight plan, the parts that need no device: entries of
    parameters that no longer exist are dropped (every 256th miss), a hit needs the same object,
    version and storage, and `restrict_pack_plan` names the parameters a capture may pack."""
    import gc

    import torch
    from segmentron_amd import functional as F
    F.clear_weight_cache()
    keep = torch.nn.Parameter(torch.ones(4, 4))
    calls = []

    def builder(p):
        def make():
            calls.append(1)
            return p.detach().clone()
        return make
    a = F.cached_pack(keep, "k", builder(keep))
    assert F.cached_pack(keep, "k", builder(keep)) is a and len(calls) == 1      # hit
    with torch.no_grad():
        keep.add_(1.0)                                                           # version bump
    b = F.cached_pack(keep, "k", builder(keep))
    assert b is not a and len(calls) == 2 and float(b[0, 0]) == 2.0
    # a discarded model's entries go away without an explicit clear
    dead = [torch.nn.Parameter(torch.zeros(8)) for _ in range(40)]
    for p in dead:
        F.cached_pack(p, "k", builder(p))
    n_before = len(F._WCACHE)
    assert n_before >= 41
    del dead, p
    gc.collect()
    for _ in range(256):  # misses on a live parameter: the purge runs at least once
        with torch.no_grad():
            keep.add_(1.0)
        F.cached_pack(keep, "k", builder(keep))
    assert len(F._WCACHE) <= 2, len(F._WCACHE)
    assert (id(keep), "k") in F._WCACHE
    # the capture-time restriction of the multi-tensor pack plan
    assert F._PLAN_FILTER[0] is None
    other = torch.nn.Parameter(torch.ones(2))
    with F.restrict_pack_plan([keep]):
        assert id(keep) in F._PLAN_FILTER[0] and id(other) not in F._PLAN_FILTER[0]
        with F.restrict_pack_plan([other]):
            assert id(other) in F._PLAN_FILTER[0] and id(keep) not in F._PLAN_FILTER[0]
        assert id(keep) in F._PLAN_FILTER[0]
    assert F._PLAN_FILTER[0] is None
    F.clear_weight_cache()


def test_fcn_head_pads_odd_hidden_width_but_keeps_the_reference_state_dict():
    """DeepLabv3+ / xception65 with SOLVER.AUX True: _FCNHead(728, nclass) has 728 // 4 = 182 hidden
    channels in the reference (deeplabv3_plus.py:29-30, module.py:13-26).  The HIP module holds 184
    (16-byte channel vectors) and speaks 182 in state_dict() / load_state_dict()."""
    import torch
    import segmentron_amd
    from segmentron_amd.config import cfg, reset_cfg
    from conftest import C3_OVERRIDES
    reset_cfg()
    cfg.update_from_list(C3_OVERRIDES + ["SOLVER.AUX", "True"])
    cfg.PHASE = "test"
    cfg.check_and_freeze()
    model = segmentron_amd.get_segmentation_model()
    sd = model.state_dict()
    want = {"auxlayer.block.0.weight": (182, 728, 3, 3), "auxlayer.block.1.weight": (182,),
            "auxlayer.block.1.bias": (182,), "auxlayer.block.1.running_mean": (182,),
            "auxlayer.block.1.running_var": (182,), "auxlayer.block.1.num_batches_tracked": (),
            "auxlayer.block.4.weight": (19, 182, 1, 1), "auxlayer.block.4.bias": (19,)}
    assert {k: tuple(v.shape) for k, v in sd.items() if k.startswith("auxlayer.")} == want
    new = {k: (torch.randn_like(v) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    model.load_state_dict(new, strict=True)
    back = model.state_dict()
    assert all(torch.equal(back[k], new[k]) for k in new)
    head = model.auxlayer
    assert head.block[0].weight.shape[0] == 184 and head.block[4].weight.shape[1] == 184
    assert head.block[0].weight[182:].abs().max().item() == 0.0
    assert head.block[4].weight[:, 182:].abs().max().item() == 0.0
    assert torch.equal(head.block[1].weight[182:].detach(), torch.ones(2))
    reset_cfg()
