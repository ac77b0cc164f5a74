"""HIP-graph execution of fixed-shape forward passes and whole train steps.

A C3 train step issues ~1230 kernels; launched one by one from Python the host is the bottleneck
(BENCH_r01: 39 ms/step for 33 ms of kernels), and the inference configs (HRNet: hundreds of small
kernels per batch) are launch-bound outright.  Every entry point of the C-ABI is capture-safe (no
allocation, no host synchronisation), so a fixed-shape pass is captured ONCE into a
`torch.cuda.CUDAGraph` (= hipGraph) and replayed.

    infer = GraphedInference(model, example_images)      # eval mode, torch.no_grad
    logits = infer(images)[0]                            # images copied into the static buffer

    step = GraphedTrainStep(model, optimizer, images, targets, loss_fn)
    loss = step()                                        # forward + loss + backward + optimizer

The static input / output tensors belong to the graph: results are overwritten by the next call.
One process per GPU, single stream.  Data-parallel steps are captured too when their exchange
steps are direct RCCL calls (`parallel.use_native_rccl`, `GraphedTrainStep(post_backward=...)`);
DistributedDataParallel / torch.distributed collectives stay eager (ProcessGroupNCCL's watchdog
aborts on events recorded in a capturing stream — measured r03).
"""
import contextlib
import gc

import torch

from . import functional as F


@contextlib.contextmanager
def capture(graph, **kw):
    """`torch.cuda.graph(graph, **kw)` with Python's cyclic garbage collector out of the way.
    A collection that runs DURING a capture may finalize a dead object that owns device resources
    — a discarded model whose TransparentTrainGraph (a reference cycle with the model) still holds
    captured graphs: destroying a hipGraph / releasing its pool inside another capture is an
    illegal call in global capture mode and the C++ destructor aborts the process (measured r04:
    two graph-mode models built one after the other).  torch >= 2.10 no longer collects on entry
    (torch.compiler.config.force_cudagraph_gc), so: collect BEFORE, keep the collector off until
    the capture has ended (reference-counted frees go through the caching allocator as usual)."""
    # (BatchNorm counter increments are owned by functional.bn_counter_scope: nothing can be
    # pending between two model forwards, so a capture neither flushes nor drops anything)
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, **kw):
            yield
    finally:
        if was_enabled:
            gc.enable()


def _warm(fn, n=3):
    """Eager warm-up on a side stream (lazy initialisation, allocator growth, weight-pack caches),
    as torch's graph-capture recipe asks."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(n):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()


class GraphedInference:
    def __init__(self, model, example, warmup=3):
        self.model = model.eval()
        self.x = example.clone()
        with torch.no_grad():
            _warm(lambda: self.model(self.x), warmup)
            F.clear_weight_cache()  # the weight packs must be issued INSIDE the capture
            self.graph = torch.cuda.CUDAGraph()
            with F.restrict_pack_plan(self.model.parameters()), capture(self.graph):
                self.out = self.model(self.x)
        torch.cuda.synchronize()

    def __call__(self, images=None):
        if images is not None:
            if images.shape != self.x.shape:
                raise ValueError("GraphedInference was captured for %s, got %s"
                                 % (tuple(self.x.shape), tuple(images.shape)))
            self.x.copy_(images)
        self.graph.replay()
        return self.out


class GraphedTrainStep:
    """forward -> loss_fn(outputs, targets) -> backward -> optimizer.step() as one graph."""

    def __init__(self, model, optimizer, images, targets, loss_fn, warmup=3, post_backward=None,
                 check=None, check_every=200):
        """post_backward: called between backward and the optimizer step — the data-parallel
        gradient averaging (`parallel.average_gradients`: direct RCCL calls on the capturing
        stream become nodes of the graph; torch.distributed collectives do NOT survive a capture
        on this stack, see segmentron_amd/rccl.py).
        check: health check of the exchange path (`xgmi.StatsExchange.check`: synchronises and
        raises when a peer stopped publishing its SyncBatchNorm statistics), called every
        `check_every` replays — a stalled peer additionally turns the loss NaN at once (p2p.h)."""
        self.images, self.targets = images, targets
        self._check, self._check_every, self._replays = check, int(check_every), 0

        def eager():
            loss = loss_fn(model(self.images), self.targets)
            optimizer.zero_grad(set_to_none=True)
            loss.backward()
            if post_backward is not None:
                post_backward()
            optimizer.step()
            return loss
        self.eager = eager
        _warm(eager, warmup)
        F.clear_weight_cache()
        optimizer.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        with F.restrict_pack_plan(model.parameters()), capture(self.graph):
            self.loss = loss_fn(model(self.images), self.targets)
            self.loss.backward()
            if post_backward is not None:
                post_backward()
            optimizer.step()
        torch.cuda.synchronize()

    def __call__(self, images=None, targets=None):
        if images is not None:
            self.images.copy_(images)
        if targets is not None:
            self.targets.copy_(targets)
        self.graph.replay()
        F.note_running_stats_changed()  # (the replayed finalize kernels rewrote running statistics)
        self._replays += 1
        if self._check is not None and self._check_every > 0 \
                and self._replays % self._check_every == 0:
            self._check()
        return self.loss

    def release(self):
        """Call before going back to eager steps: the optimizer stepped inside the graph without
        moving `param._version`, so the cached weight packs are stale."""
        F.clear_weight_cache()


# ----------------------------------------------------------------------------- transparent capture
class _GraphedSegment(torch.autograd.Function):
    """One captured forward / backward pair as an autograd node (the torch.cuda
    make_graphed_callables split): forward replays the forward graph and hands out the static
    low-resolution logits; backward copies their gradients into the static buffers, replays
    the backward graph and ASSIGNS the static parameter gradients (`p.grad = g`, no
    AccumulateGrad clone of 440 tensors; an existing different gradient is added to)."""

    @staticmethod
    def forward(ctx, anchor, seg):
        seg.fwd.replay()
        F.note_running_stats_changed()  # (functional.eval_affine: cached evaluation-mode affines)
        ctx.seg = seg
        return tuple(t.detach() for t in seg.lo)

    @staticmethod
    def backward(ctx, *grads):
        seg = ctx.seg
        for buf, g in zip(seg.grad_lo, grads):
            if g is None:
                buf.zero_()
            else:
                buf.copy_(g)
        with torch.no_grad():
            # `p.grad` still IS the static buffer of an earlier backward: the caller accumulates
            # gradients over several backward calls (or scaled / clipped / zeroed them in place —
            # a version counter cannot tell these apart, so nothing is inferred from it): keep
            # the current values, the replay overwrites the buffers, add them back.  The
            # reference's loop (zero_grad() -> set_to_none) never takes this path: it is the
            # fast path; a loop on zero_grad(set_to_none=False) pays a full-model gradient clone
            # and a foreach_add per step here.
            held = [g for p, g in zip(seg.params, seg.grads) if g is not None and p.grad is g]
            old = [g.clone() for g in held]
            seg.bwd.replay()
            if held:
                torch._foreach_add_(held, old)
            for p, g in zip(seg.params, seg.grads):
                if g is None:
                    continue
                if p.grad is None or p.grad is g:
                    p.grad = g
                else:  # gradient accumulation / zero_grad(set_to_none=False) on another tensor
                    p.grad.add_(g)
        return None, None


class _Segment:
    __slots__ = ("fwd", "bwd", "x", "lo", "meta", "grad_lo", "params", "grads", "anchor",
                 "container")


class TransparentTrainGraph:
    """HIP-graph execution behind the reference's UNCHANGED train loop (tools/train.py:130-146):

        outputs = self.model(images); loss_dict = self.criterion(outputs, targets); ...
        self.optimizer.zero_grad(); losses.backward(); self.optimizer.step()

    `install(model)` (done by `get_segmentation_model()` when SEGMENTRON_HIP_GRAPH=1) replaces
    `model.forward`: in training mode with gradients enabled, the first `warmup` calls with a
    given input shape run eagerly (they are real steps: allocator, lazy initialisation, weight
    packs warm up on them); the next call captures TWO graphs keyed on that shape — the forward
    pass up to the heads' low-resolution logits, and its backward pass down to the parameter
    gradients — and from then on `model(images)` is one copy + one graph replay and
    `losses.backward()` one replay.  The criterion (the fused upsample + cross-entropy kernels,
    through `functional.LogitsView`) and `optimizer.step()` (FusedSGD: 11 launches) stay eager
    launches: ~15 of the ~1140 launches of a step.  EVALUATION-mode forwards under
    `torch.no_grad()` (tools/eval.py:70-78 -> `SegBaseModel.evaluate` -> `self.forward`, and the
    validation pass of tools/train.py:170-190) are captured the same way, one forward graph per
    input shape (up to `max_eval_shapes`: multi-scale testing feeds a few), and return COPIES of
    the static outputs (`evaluate` adds the flipped pass to the unflipped one, segbase.py:69-72).
    Any other call — gradients enabled in evaluation mode, a new shape beyond the limits, a
    model whose BatchNorms synchronise over torch.distributed — takes the eager path, which
    stays correct; parameters may be updated by any optimizer (weight packs are re-issued inside
    the captured forward).  One process per GPU, one stream.

    What a loop may rely on (tests/test_train_loop_gpu.py, all bit-identical to eager launches):
    gradient accumulation over several forward/backward pairs without zero_grad; training,
    validation (model.eval() under no_grad) and training again on the same object; every model
    of the zoo (LogitsView outputs, HRNet's materialised tensor, lists and tuples, aux heads).
    What it may NOT do, as with torch.cuda.make_graphed_callables: keep a training-mode output
    across the next training forward of the same shape (static tensors), or run two forwards
    before their two backwards."""

    def __init__(self, model, warmup=2, max_shapes=2, max_eval_shapes=6):
        self.model, self.warmup, self.max_shapes = model, int(warmup), int(max_shapes)
        self.max_eval_shapes = int(max_eval_shapes)
        self.eager_forward = model.forward
        self.seen, self.segments, self.disabled = {}, {}, None
        self.eval_segments = {}

    # -- installation
    @classmethod
    def install(cls, model, **kw):
        tg = cls(model, **kw)
        model.forward = tg.forward          # nn.Module.__call__ -> self.forward
        model._transparent_graph = tg
        return tg

    def uninstall(self):
        self.model.__dict__.pop("forward", None)
        self.model.__dict__.pop("_transparent_graph", None)
        self.segments.clear()
        self.eval_segments.clear()

    # -- dispatch
    def _key(self, x):
        return (tuple(x.shape), x.dtype, x.device.index)

    def _capturable(self, x):
        import torch.distributed as dist
        if not (self.model.training and torch.is_grad_enabled() and isinstance(x, torch.Tensor)
                and x.is_cuda and not x.requires_grad and self.disabled is None):
            return False
        if torch.cuda.is_current_stream_capturing():
            return False  # somebody else (GraphedTrainStep) is capturing the whole step
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from . import parallel
            if parallel.native_rccl() is None:
                return False  # torch.distributed collectives do not survive a capture
        return True

    def _eval_capturable(self, x):
        return (not self.model.training and not torch.is_grad_enabled()
                and isinstance(x, torch.Tensor) and x.is_cuda and self.disabled is None
                and not torch.cuda.is_current_stream_capturing()
                and not F.lazy_eval_logits())

    def _eval_forward(self, x):
        key = self._key(x)
        seg = self.eval_segments.get(key)
        if seg is None:
            n = self.seen.get(("eval",) + key, 0)
            self.seen[("eval",) + key] = n + 1
            if n < self.warmup or len(self.eval_segments) >= self.max_eval_shapes:
                return self.eager_forward(x)
            try:
                seg = _Segment()
                seg.x = x.clone()
                torch.cuda.synchronize()
                F.clear_weight_cache()  # packs inside the graph: replays follow the optimizer
                seg.fwd = torch.cuda.CUDAGraph()
                with F.restrict_pack_plan(self.model.parameters()), capture(seg.fwd):
                    outs = self.eager_forward(seg.x)
                if not all(isinstance(o, torch.Tensor) for o in outs):
                    raise RuntimeError("evaluation capture needs tensor outputs")
                seg.lo = list(outs)
                torch.cuda.synchronize()
                F.clear_weight_cache()
            except Exception as e:  # noqa: BLE001
                import sys
                self.disabled = repr(e)[:300]
                sys.stderr.write("segmentron_amd.graph: evaluation capture failed (%s): eager "
                                 "launches from here on\n" % self.disabled)
                torch.cuda.synchronize()
                F.clear_weight_cache()
                return self.eager_forward(x)
            self.eval_segments[key] = seg
        seg.x.copy_(x)
        seg.fwd.replay()
        return tuple(t.clone() for t in seg.lo)

    def forward(self, x, *args, **kwargs):
        if not args and not kwargs and self._eval_capturable(x):
            return self._eval_forward(x)
        if args or kwargs or not self._capturable(x):
            return self.eager_forward(x, *args, **kwargs)
        key = self._key(x)
        seg = self.segments.get(key)
        if seg is None:
            n = self.seen.get(key, 0)
            self.seen[key] = n + 1
            if n < self.warmup or len(self.segments) >= self.max_shapes:
                return self.eager_forward(x)
            try:
                seg = self._capture(x)
            except Exception as e:  # noqa: BLE001 — eager stays correct; say why, once
                import sys
                self.disabled = repr(e)[:300]
                sys.stderr.write("segmentron_amd.graph: transparent capture failed (%s): eager "
                                 "launches from here on\n" % self.disabled)
                torch.cuda.synchronize()
                F.clear_weight_cache()
                return self.eager_forward(x)
            self.segments[key] = seg
        seg.x.copy_(x)
        lo = _GraphedSegment.apply(seg.anchor, seg)
        return seg.container(t if m is None else F.LogitsView(t, m[0], m[1])
                             for t, m in zip(lo, seg.meta))

    # -- capture of one shape
    def _capture(self, x):
        model = self.model
        seg = _Segment()
        seg.x = x.clone()
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        seg.params = [p for _, p in named]
        # The captured passes run on fresh LEAF ALIASES of the parameters (same storage, same
        # version counter).  A parameter's AccumulateGrad node is created once and stays bound to
        # the stream it was created on for as long as any autograd graph references it — in the
        # reference loop the previous iteration's `losses` still does at this point — and the
        # engine then makes THAT stream (the legacy default stream) wait for the capturing one:
        # the default stream is dragged into the capture, never re-joined, and hipStreamEndCapture
        # dies (measured r04, AMD_LOG_LEVEL=3: `hipStreamWaitEvent(stream:<null>, ...)` inside the
        # capture).  The aliases get their gradient edges created inside the capture.
        leaves = [p.detach().requires_grad_(True) for p in seg.params]
        seg.anchor = torch.zeros((), device=x.device, requires_grad=True)
        torch.cuda.synchronize()
        F.clear_weight_cache()  # the weight packs must be issued INSIDE the captured forward
        # Two PRIVATE memory pools.  With one shared pool (the make_graphed_callables recipe) a
        # parameter gradient may be placed in memory a forward activation occupied until the
        # backward pass freed it: `p.grad` (the static buffer itself) would then be destroyed by
        # the NEXT forward replay — fine for forward/backward/step, wrong as soon as gradients
        # are accumulated over several forward/backward pairs (measured r04: 339 of 440 tensors).
        # Memory of the backward pool is only ever written by backward replays.
        seg.fwd, seg.bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # (the aliases resolve to their parameters in functional.packed_pointwise: one
        # multi-tensor pack launch for the whole model instead of one per alias)
        F._PARAM_ALIAS.update({id(a): p for a, p in zip(leaves, seg.params)})
        try:
            with F.restrict_pack_plan(seg.params), capture(seg.fwd):
                outs = torch.func.functional_call(
                    model, {n: t for (n, _), t in zip(named, leaves)}, (seg.x,))
            # outputs: LogitsView (low-resolution logits, upsampled inside the fused loss) or a
            # materialised tensor (HRNet's align_corners=False boundary)
            if not all(isinstance(o, (F.LogitsView, torch.Tensor)) for o in outs):
                raise RuntimeError("transparent capture needs tensor / LogitsView outputs")
            seg.container = list if isinstance(outs, list) else tuple
            seg.lo = [o.lo if isinstance(o, F.LogitsView) else o for o in outs]
            seg.meta = [(o.out_hw, o.align_corners) if isinstance(o, F.LogitsView) else None
                        for o in outs]
            seg.grad_lo = [torch.zeros_like(t) for t in seg.lo]
            with F.restrict_pack_plan(seg.params), capture(seg.bwd):
                grads = torch.autograd.grad(seg.lo, leaves, seg.grad_lo, allow_unused=True)
            # (a .contiguous() copy taken here would NOT be a node of the graph: replays would
            # never refresh it)
            ragged = [n for (n, _), g in zip(named, grads) if g is not None and not g.is_contiguous()]
            if ragged:  # (named: the eager fallback this triggers must be diagnosable)
                raise RuntimeError("transparent capture needs contiguous parameter gradients; "
                                   "not contiguous: " + ", ".join(ragged[:4]))
            seg.grads = list(grads)
            torch.cuda.synchronize()
        finally:  # (ids of dead aliases would be recycled by unrelated tensors)
            for a in leaves:
                F._PARAM_ALIAS.pop(id(a), None)
            F.clear_weight_cache()  # (entries keyed on the aliases)
        return seg


def transparent_graph_requested():
    import os
    return os.environ.get("SEGMENTRON_HIP_GRAPH", "0") == "1"
