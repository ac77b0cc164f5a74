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
import torch

from . import functional as F


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
            with torch.cuda.graph(self.graph):
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
        with torch.cuda.graph(self.graph):
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
        self._replays += 1
        if self._check is not None and self._check_every > 0 \
                and self._replays % self._check_every == 0:
            self._check()
        return self.loss

    def release(self):
        """Call before going back to eager steps: the optimizer stepped inside the graph without
        moving `param._version`, so the cached weight packs are stale."""
        F.clear_weight_cache()
