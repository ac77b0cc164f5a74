"""The reference's UNCHANGED training-loop statements (tools/train.py:135-146) on the HIP path,
eager launches vs SEGMENTRON_HIP_GRAPH=1 (segmentron_amd.graph.TransparentTrainGraph: forward and
backward replay captured HIP graphs behind `model(images)` / `losses.backward()`).

    outputs = self.model(images)
    loss_dict = self.criterion(outputs, targets)
    losses = sum(loss for loss in loss_dict.values())
    self.optimizer.zero_grad()
    losses.backward()
    self.optimizer.step()
    self.lr_scheduler.step()

Same kernels on the same operands in the same order, fixed-order reductions -> the two modes must
agree BIT FOR BIT: every loss, every parameter, every BatchNorm buffer after 7 iterations with a
different batch each (2 eager warm-up calls, the capturing call, 4 replays), with the optimizer
and scheduler `segmentron.solver` hands to tools/train.py and a criterion of the reference's
MixSoftmaxCrossEntropyLoss shape (solver/loss.py:16-46: an nn.CrossEntropyLoss subclass that
returns dict(loss=...))."""
import os

import pytest
import torch
import torch.nn as nn

from conftest import C3_OVERRIDES
from oracle import synth

pytestmark = pytest.mark.gpu


class MixSoftmaxCrossEntropyLoss(nn.CrossEntropyLoss):
    """Shape of segmentron/solver/loss.py:16-46 (aux outputs weighted by aux_weight)."""

    def __init__(self, aux=False, aux_weight=0.4, ignore_index=-1):
        super().__init__(ignore_index=ignore_index)
        self.aux, self.aux_weight = aux, aux_weight

    def forward(self, *inputs, **kwargs):
        preds, target = tuple(inputs)
        loss = super().forward(preds[0], target)
        for p in preds[1:]:
            loss = loss + self.aux_weight * super().forward(p, target)
        return dict(loss=loss)


def _run_loop(graph, iters, hw, dtype=torch.bfloat16, micro=1):
    import segmentron_amd
    from segmentron_amd.config import cfg, reset_cfg
    from segmentron_amd.solver.lr_scheduler import get_scheduler
    from segmentron_amd.solver.optimizer import get_optimizer
    prev = os.environ.get("SEGMENTRON_HIP_GRAPH")
    os.environ["SEGMENTRON_HIP_GRAPH"] = "1" if graph else "0"
    try:
        reset_cfg()
        # (SOLVER.AUX stays False: DeepLabv3+'s aux head on xception is a 728 -> 182-channel
        # _FCNHead, and 182 is not a multiple of the kernels' 16-byte channel vector)
        cfg.update_from_list(C3_OVERRIDES + ["SOLVER.LR", "0.002"])
        cfg.PHASE = "train"
        cfg.check_and_freeze()
        segmentron_amd.set_compute_dtype(dtype)
        model = segmentron_amd.get_segmentation_model()
        assert (getattr(model, "_transparent_graph", None) is not None) == graph
        sd = synth.synth_like(model.state_dict(), seed=0, conditioned=True)
        model.load_state_dict(sd)
        model = model.to("cuda")
        criterion = MixSoftmaxCrossEntropyLoss(aux=False, aux_weight=cfg.SOLVER.AUX_WEIGHT,
                                               ignore_index=cfg.DATASET.IGNORE_INDEX).to("cuda")
        optimizer = get_optimizer(model)
        lr_scheduler = get_scheduler(optimizer, max_iters=iters, iters_per_epoch=iters)
        model.train()
        for m in model.modules():  # RNG-free: graph replays draw their masks differently
            if isinstance(m, (nn.Dropout, nn.Dropout2d)):
                m.p = 0.0
        H, W = hw
        losses_seen = []
        for it in range(iters):
            if micro > 1:
                # gradient accumulation over `micro` batches (not in the reference's loop; a
                # common edit of it): zero_grad once, several forward/backward pairs, one step
                optimizer.zero_grad()
                for k in range(micro):
                    images = synth.synth_images(2, H, W, seed=100 + it * micro + k).to("cuda")
                    targets = synth.synth_targets(2, H, W, seed=100 + it * micro + k).to("cuda")
                    outputs = model(images)
                    loss_dict = criterion(outputs, targets)
                    losses = sum(loss for loss in loss_dict.values())
                    losses.backward()
                    losses_seen.append(losses.item())
                optimizer.step()
                lr_scheduler.step()
                continue
            images = synth.synth_images(2, H, W, seed=100 + it).to("cuda")
            targets = synth.synth_targets(2, H, W, seed=100 + it).to("cuda")
            # ---- tools/train.py:135-146, verbatim
            outputs = model(images)
            loss_dict = criterion(outputs, targets)
            losses = sum(loss for loss in loss_dict.values())
            optimizer.zero_grad()
            losses.backward()
            optimizer.step()
            lr_scheduler.step()
            # ----
            losses_seen.append(losses.item())
        torch.cuda.synchronize()
        tg = getattr(model, "_transparent_graph", None)
        if graph:
            assert tg.disabled is None, tg.disabled
            assert len(tg.segments) == 1, "the loop did not reach the captured path"
        state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        lrs = [g["lr"] for g in optimizer.param_groups]
        # an evaluation pass on the same object still works (eager, eval mode, no_grad)
        model.eval()
        with torch.no_grad():
            ev = model(synth.synth_images(1, H, W, seed=5).to("cuda"))
        assert isinstance(ev[0], torch.Tensor) and ev[0].shape == (1, 19, H, W)
        return losses_seen, state, lrs
    finally:
        if prev is None:
            os.environ.pop("SEGMENTRON_HIP_GRAPH", None)
        else:
            os.environ["SEGMENTRON_HIP_GRAPH"] = prev
        reset_cfg()


def test_reference_loop_statements_graph_mode_equals_eager_bit_for_bit():
    iters, hw = 7, (129, 193)
    le, se, lre = _run_loop(False, iters, hw)
    lg, sg, lrg = _run_loop(True, iters, hw)
    print("train-loop losses eager %s\n           graph %s" % (["%.5f" % v for v in le],
                                                                ["%.5f" % v for v in lg]))
    assert le == lg
    assert lre == lrg
    bad = [k for k in se if not torch.equal(se[k], sg[k])]
    assert not bad, bad[:5]
    assert le[-1] < le[0]  # the loop trains


def test_gradient_accumulation_graph_mode_equals_eager_bit_for_bit():
    """Three forward/backward pairs per optimizer step, no zero_grad in between: the first
    iteration mixes eager backward passes with the capturing one (the static gradients are ADDED
    to the accumulated tensors), the later ones accumulate INTO the static buffers (the replay
    overwrites them: _GraphedSegment.backward keeps the old values when nobody has written to
    `p.grad` since the previous backward)."""
    iters, hw = 3, (65, 129)
    le, se, _ = _run_loop(False, iters, hw, micro=3)
    lg, sg, _ = _run_loop(True, iters, hw, micro=3)
    assert le == lg
    bad = [k for k in se if not torch.equal(se[k], sg[k])]
    assert not bad, bad[:5]


def test_evaluate_graph_mode_equals_eager_bit_for_bit():
    """tools/eval.py:70-78 / tools/train.py:170-190: `model.evaluate(image)` (multi-scale, flip,
    pad — segbase.py:44-79) with SEGMENTRON_HIP_GRAPH=1: every evaluation-mode forward of a shape
    replays a captured graph from its third call on (copies of the static outputs are returned:
    evaluate() adds the flipped pass to the unflipped one).  Bit-identical to eager launches."""
    import test_model_gpu as M
    extra = ["TEST.SCALES", "[0.75, 1.0]", "TEST.FLIP", "True", "TEST.CROP_SIZE", "(81, 145)"]
    xs = [synth.synth_images(1, 65, 129, seed=40 + i).cuda() for i in range(5)]

    def run(graph):
        prev = os.environ.get("SEGMENTRON_HIP_GRAPH")
        os.environ["SEGMENTRON_HIP_GRAPH"] = "1" if graph else "0"
        try:
            model, _ = M._build_eval(extra)
            tg = getattr(model, "_transparent_graph", None)
            assert (tg is not None) == graph
            with torch.no_grad():
                outs = [model.evaluate(x).clone() for x in xs]
            if graph:
                assert tg.disabled is None, tg.disabled
                assert len(tg.eval_segments) == 2 and not tg.segments  # the two scales' shapes
            return outs
        finally:
            if prev is None:
                os.environ.pop("SEGMENTRON_HIP_GRAPH", None)
            else:
                os.environ["SEGMENTRON_HIP_GRAPH"] = prev

    a, b = run(False), run(True)
    for i, (u, v) in enumerate(zip(a, b)):
        assert torch.equal(u, v), i
    assert not torch.equal(a[0], a[1])


def _train_validate_train(tag, graph, hw, epochs=2, iters=4, builder=None):
    """tools/train.py's epoch structure: `iters` loop iterations, then `validation()`
    (train.py:173-199: model.eval(), `model(image)[0]` under no_grad per sample), `model.train()`,
    and on."""
    import segmentron_amd
    import test_more_models as MM
    from segmentron_amd.config import cfg, reset_cfg
    from segmentron_amd.solver.lr_scheduler import get_scheduler
    from segmentron_amd.solver.optimizer import get_optimizer
    prev = os.environ.get("SEGMENTRON_HIP_GRAPH")
    os.environ["SEGMENTRON_HIP_GRAPH"] = "1" if graph else "0"
    try:
        model = builder() if builder is not None else MM._build_hip(tag, torch.bfloat16, True)[0]
        tg = getattr(model, "_transparent_graph", None)
        assert (tg is not None) == graph
        criterion = MixSoftmaxCrossEntropyLoss(aux=True, aux_weight=0.4, ignore_index=-1).cuda()
        optimizer = get_optimizer(model)
        lr_scheduler = get_scheduler(optimizer, max_iters=epochs * iters, iters_per_epoch=iters)
        H, W = hw
        losses_seen, val_seen = [], []
        for ep in range(epochs):
            for it in range(iters):
                images = synth.synth_images(2, H, W, seed=300 + ep * iters + it).cuda()
                targets = synth.synth_targets(2, H, W, seed=300 + ep * iters + it).cuda()
                outputs = model(images)
                loss_dict = criterion(outputs, targets)
                losses = sum(loss for loss in loss_dict.values())
                optimizer.zero_grad()
                losses.backward()
                optimizer.step()
                lr_scheduler.step()
                losses_seen.append(losses.item())
            model.eval()
            for i in range(3):
                image = synth.synth_images(1, H, W, seed=900 + i).cuda()
                with torch.no_grad():
                    output = model(image)[0]
                val_seen.append(output.float().clone())
            model.train()
        torch.cuda.synchronize()
        if graph:
            assert tg.disabled is None, tg.disabled
            assert len(tg.segments) == 1 and len(tg.eval_segments) == 1, \
                (len(tg.segments), len(tg.eval_segments))
        state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        # the KNOWN iteration count, not only eager == graph (r04: both sides were equally wrong):
        # every BatchNorm that runs in a training forward has counted every one of them
        counters = {k: int(v) for k, v in state.items() if k.endswith("num_batches_tracked")}
        assert counters and set(counters.values()) <= {0, epochs * iters}, \
            sorted(set(counters.values()))
        assert sum(v == epochs * iters for v in counters.values()) >= len(counters) // 2
        from segmentron_amd import functional as HF
        assert not HF._PENDING_COUNTERS and HF._COUNTER_SCOPE[0] == 0
        return losses_seen, val_seen, state
    finally:
        if prev is None:
            os.environ.pop("SEGMENTRON_HIP_GRAPH", None)
        else:
            os.environ["SEGMENTRON_HIP_GRAPH"] = prev
        reset_cfg()


@pytest.mark.parametrize("tag", ["c4", "c5", "c7", "c8", "c6"])
def test_train_validate_train_graph_mode_equals_eager_bit_for_bit(tag):
    """PSPNet (aux head), HRNet, Fast-SCNN (two aux heads), DANet (three outputs), CCNet through
    the epoch structure of tools/train.py with SEGMENTRON_HIP_GRAPH=1: the training graphs keep
    replaying after the evaluation-mode graph of the validation pass was captured between them,
    validation reads the running statistics the training replays wrote.  Bit-identical to eager
    launches: every loss, every validation output, every parameter and buffer."""
    import test_more_models as MM
    hw = MM.CASES[tag]["hw"]
    le, ve, se = _train_validate_train(tag, False, hw)
    lg, vg, sg = _train_validate_train(tag, True, hw)
    assert le == lg, (le, lg)
    for i, (a, b) in enumerate(zip(ve, vg)):
        assert torch.equal(a, b), ("validation output", i)
    bad = [k for k in se if not torch.equal(se[k], sg[k])]
    assert not bad, bad[:5]
    assert not torch.equal(ve[0], ve[3])  # the second epoch validates an updated model


def _build_pspnet_group_norm():
    """PSPNet / resnet50, cfg.MODEL.BN_TYPE 'GN' (GroupNorm in the encoder, BatchNorm in the
    heads as in the reference), auxiliary head, bf16, conditioned synthetic state."""
    import segmentron_amd
    from segmentron_amd.config import cfg, reset_cfg
    reset_cfg()
    cfg.update_from_list(["DATASET.NAME", "cityscape", "MODEL.MODEL_NAME", "PSPNet",
                          "MODEL.BACKBONE", "resnet50", "MODEL.OUTPUT_STRIDE", "8",
                          "MODEL.BN_TYPE", "GN", "SOLVER.AUX", "True", "SOLVER.LR", "0.002",
                          "TRAIN.BACKBONE_PRETRAINED", "False"])
    cfg.PHASE = "train"
    cfg.check_and_freeze()
    segmentron_amd.set_compute_dtype(torch.bfloat16)
    model = segmentron_amd.get_segmentation_model()
    model.load_state_dict(synth.synth_like(model.state_dict(), seed=2, conditioned=True))
    model = model.cuda().train()
    for m in model.modules():
        if isinstance(m, (nn.Dropout, nn.Dropout2d)):
            m.p = 0.0
    return model


def test_group_norm_model_graph_mode_equals_eager_bit_for_bit():
    """The GroupNorm kernels (csrc/groupnorm.hip: moments / finalize / affine, both directions) are
    capture-safe: the train / validate / train structure of tools/train.py on PSPNet-resnet50 with
    BN_TYPE 'GN' under SEGMENTRON_HIP_GRAPH=1 equals eager launches bit for bit."""
    hw = (65, 97)
    le, ve, se = _train_validate_train("gn", False, hw, epochs=2, iters=3, builder=_build_pspnet_group_norm)
    lg, vg, sg = _train_validate_train("gn", True, hw, epochs=2, iters=3, builder=_build_pspnet_group_norm)
    assert le == lg, (le, lg)
    assert all(l == l and abs(l) < 1e4 for l in le), le
    for i, (a, b) in enumerate(zip(ve, vg)):
        assert torch.equal(a, b), ("validation output", i)
    bad = [k for k in se if not torch.equal(se[k], sg[k])]
    assert not bad, bad[:5]
