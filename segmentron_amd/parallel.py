"""Data-parallel exchange steps of the hot path (one process per GPU, RCCL over xGMI through
torch.distributed backend "nccl"; the same code runs on gloo for the CPU tests).

The path shards by image.  Two exchange steps exist per training step:
  * gradients: DistributedDataParallel bucketed all-reduce (torch, unchanged — tools/train.py:108-111)
  * SyncBatchNorm statistics (tools/train.py:76): per BN ONE all-reduce of the 2C float64 sums
    [sum x, sum x^2] in forward and ONE of 2C [sum g', sum g'*x] (or [ds, dt] for folded layers)
    in backward — torch's nn.SyncBatchNorm instead all_gathers (mean, invstd, count) in forward
    (torch/nn/modules/_functions.py:49,74).  Same statistics, fewer and smaller messages.
"""
import torch.distributed as dist
import torch.nn as nn


def _FORCE():
    """SEG_SYNC_FORCE=1 (test plumbing, set by bench.py's SEG_BENCH_FORCE_DDP): run the SyncBN
    exchange with a single rank too, so that a one-GPU box exercises the RCCL calls."""
    import os
    return os.environ.get("SEG_SYNC_FORCE") == "1"


_NATIVE = [None]


def use_native_rccl(comm):
    """Route the SyncBatchNorm statistics exchanges (and `average_gradients`) through a
    `segmentron_amd.rccl.Communicator` instead of torch.distributed: direct `ncclAllReduce` calls
    on the current stream — a few microseconds of host work each and, unlike ProcessGroupNCCL,
    safe inside a HIP-graph capture.  None switches back.  Returns the previous setting."""
    prev, _NATIVE[0] = _NATIVE[0], comm
    return prev


def native_rccl():
    return _NATIVE[0]


def mailbox(group=None):
    """The xGMI peer mailbox of the active exchange (xgmi.StatsExchange), or None: with it the
    BatchNorm finalize kernels exchange their sums themselves (hip_ops.*_sync) and a SyncBatchNorm
    costs the same one launch per direction as a plain BatchNorm.  The mailbox spans the ranks it
    was connected over (the default group): a BatchNorm that synchronises over a SUBGROUP
    (nn.SyncBatchNorm(process_group=...)) keeps the all-reduce path."""
    box = getattr(_NATIVE[0], "mailbox", None)
    if box is not None and not _is_world(group):
        return None
    return box


def _is_world(group):
    """Does `group` span the ranks the native communicator / mailbox was built over?"""
    if group is None or not dist.is_initialized() or group is dist.group.WORLD:
        return True
    comm = _NATIVE[0]
    return dist.get_world_size(group) == getattr(comm, "world", dist.get_world_size())


def _all_reduce(t, group):
    """The native communicator spans the default group only: a SyncBatchNorm over a SUBGROUP
    (nn.SyncBatchNorm(process_group=...)) goes through torch.distributed with its group — eager
    launches only (ProcessGroupNCCL does not survive a HIP-graph capture on this stack)."""
    if _NATIVE[0] is not None and _is_world(group):
        _NATIVE[0].all_reduce(t)
        return
    if _NATIVE[0] is not None:
        import torch
        if t.is_cuda and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("SyncBatchNorm over a process subgroup cannot be captured into a "
                               "HIP graph: the native RCCL communicator spans the default group "
                               "only (run this model with eager launches)")
    dist.all_reduce(t, group=group)


def average_gradients(params):
    """What DistributedDataParallel does for the gradients (tools/train.py:108-111: mean over
    ranks), as ONE grouped RCCL all-reduce over the native communicator after backward —
    capturable with the rest of the step."""
    comm = _NATIVE[0]
    if comm is None:
        raise RuntimeError("average_gradients needs parallel.use_native_rccl(comm)")
    grads = []
    for p in params:
        if p.grad is not None:
            if not p.grad.is_contiguous():
                p.grad = p.grad.contiguous()
            grads.append(p.grad)
    comm.all_reduce_many(grads, "avg")


class OverlappedGradientAverager:
    """DistributedDataParallel's bucketed, backward-overlapped gradient averaging
    (tools/train.py:108-111) on direct RCCL calls: parameters are bucketed in REVERSE order (the
    order backward produces their gradients, ~`bucket_bytes` each); when the last gradient of a
    bucket has been accumulated (post-accumulate-grad hooks) a side HIP stream waits for the
    compute stream and issues ONE grouped `avg` all-reduce for the bucket, while backward goes
    on; `finish()` (call it between backward and the optimizer step) makes the compute stream
    wait for the side stream.

    `comm` must be a communicator of its OWN — the SyncBatchNorm exchanges keep running on the
    compute stream over `native_rccl()`, and one communicator must not be driven from two
    streams at once.  Inside a HIP-graph capture the side stream joins the capture through the
    stream waits, so the overlap is part of the replayed graph."""

    def __init__(self, params, comm, bucket_bytes=32 << 20):
        import torch
        self.comm = comm
        self.side = torch.cuda.Stream()
        self.buckets, cur, size = [], [], 0
        for p in reversed([p for p in params if p.requires_grad]):
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self._bucket_of = {}
        for bi, b in enumerate(self.buckets):
            for p in b:
                self._bucket_of[id(p)] = bi
        self._left = [len(b) for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._paused = False
        self._hooks = [p.register_post_accumulate_grad_hook(self._ready) for b in self.buckets
                       for p in b]

    def _ready(self, p):
        bi = self._bucket_of[id(p)]
        if self._launched[bi] or self._left[bi] <= 0:
            # a second backward before finish(): the bucket's all-reduce already ran on partial
            # sums (gradient accumulation needs `with averager.no_sync():` around all but the
            # last backward, like DistributedDataParallel.no_sync)
            raise RuntimeError("OverlappedGradientAverager: a gradient arrived after its bucket "
                               "was averaged — call finish() after every backward, or wrap the "
                               "accumulating backwards in no_sync()")
        if self._paused:
            return
        self._left[bi] -= 1
        if self._left[bi] == 0:
            self._launch(bi)

    def no_sync(self):
        """Context manager: backwards inside it only accumulate (no all-reduce); the first
        backward outside it averages the accumulated gradients."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            prev, self._paused = self._paused, True
            try:
                yield
            finally:
                self._paused = prev
        return ctx()

    def _launch(self, bi):
        import torch
        grads = []
        for p in self.buckets[bi]:
            if p.grad is not None:
                if not p.grad.is_contiguous():
                    p.grad = p.grad.contiguous()
                grads.append(p.grad)
        self._launched[bi] = True
        if not grads:
            return
        main = torch.cuda.current_stream()
        self.side.wait_stream(main)  # the bucket's gradients are complete on the compute stream
        with torch.cuda.stream(self.side):
            self.comm.all_reduce_many(grads, "avg")

    def finish(self):
        """After backward: buckets whose hooks did not all fire (parameters without a gradient
        this step) are averaged now; the compute stream then waits for the side stream."""
        import torch
        for bi in range(len(self.buckets)):
            if not self._launched[bi]:
                self._launch(bi)
        torch.cuda.current_stream().wait_stream(self.side)
        self._left = [len(b) for b in self.buckets]
        self._launched = [False] * len(self.buckets)

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def is_naive_sync(bn):
    """The reference's own NaiveSyncBatchNorm (an nn.BatchNorm2d subclass)."""
    return type(bn).__name__ == "NaiveSyncBatchNorm" and isinstance(bn, nn.BatchNorm2d)


def sync_group(bn):
    """Process group to synchronise over, or None (plain BN / eval / single process)."""
    if not (bn.training and dist.is_available() and dist.is_initialized()
            and (dist.get_world_size() > 1 or _FORCE())):
        return None
    if isinstance(bn, nn.SyncBatchNorm):
        return bn.process_group if bn.process_group is not None else dist.group.WORLD
    if is_naive_sync(bn):
        return dist.group.WORLD
    return None


def naive_running_update(bn, mean, invstd):
    """Running statistics of the reference's NaiveSyncBatchNorm (batch_norm.py:174-176):
    `running += momentum * (batch - running)` with the BIASED batch variance; the counter
    `num_batches_tracked` is not touched."""
    import torch
    with torch.no_grad():
        m = bn.momentum
        var = invstd.double().pow(-2) - bn.eps  # invstd = rsqrt(var_biased + eps)
        bn.running_mean += m * (mean.to(bn.running_mean.dtype) - bn.running_mean)
        bn.running_var += m * (var.to(bn.running_var.dtype) - bn.running_var)


def allreduce_forward_sums(partial2d, local_count, group):
    """partial2d: fp32 [R, 2C] local partial rows of (sum x, sum x^2) -> (global float64 sums
    [2C], global count).  The local element count rides in the same message ([2C+1], assembled
    by ONE launch: hip_ops.colsum_count), so ranks with different N*H*W (uneven last batch,
    variable-size inputs) still get the exact global statistics — torch's SyncBatchNorm
    all-gathers per-rank counts for the same reason (torch/nn/modules/_functions.py:49-74)."""
    import torch
    if partial2d.dim() == 1:  # already summed (host-side protocol tests over gloo)
        n = partial2d.numel()
        buf = torch.empty(n + 1, dtype=partial2d.dtype, device=partial2d.device)
        buf[:n] = partial2d
        buf[n:].fill_(float(local_count))  # (a fill kernel: `buf[n] = x` is a host-to-device
        #                                    copy, which a HIP-graph capture does not allow)
    else:
        from . import hip_ops as K
        n = partial2d.shape[1]
        buf = K.colsum_count(partial2d, float(local_count))
    _all_reduce(buf, group)
    return buf[:n], buf[n:]


def allreduce_moments(buf, group):
    """buf: float64 [2C + 1] = this rank's (n*mean | M2 + n*mean^2 | n) of a SMALL BatchNorm
    (hip_ops.bn_moments_small) -> (global sums [2C], global count [1]), in place."""
    _all_reduce(buf, group)
    n = buf.numel() - 1
    return buf[:n], buf[n:]


def allreduce_backward_sums(sums, group):
    """sums: [2C] local (sum g', sum g'*x) or (ds, dt) -> global, in place."""
    _all_reduce(sums, group)
    return sums


def grad_scale(group):
    """Every rank computes dgamma / dbeta from GLOBAL sums; the data-parallel gradient averaging
    (DistributedDataParallel / average_gradients) divides by the world size once more, and
    torch's SyncBatchNorm returns LOCAL sums there — the finalize kernels multiply by this
    factor so that the averaged value reproduces the reference's."""
    return 1.0 / dist.get_world_size(group)


def local_param_grads(dgamma, dbeta, group):
    """The same as two tensor divisions (paths whose finalize kernel takes no scale)."""
    ws = dist.get_world_size(group)
    return dgamma / ws, dbeta / ws
