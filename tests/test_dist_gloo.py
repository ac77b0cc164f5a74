"""CPU, world_size 2, gloo: the data-parallel exchange protocol of the hot path
(segmentron_amd/parallel.py — the code the HIP path calls under SyncBatchNorm + DDP).
Each rank holds half of a batch; after the exchanges the BatchNorm forward statistics, the
input gradient of its shard and the DDP-averaged parameter gradients must equal single-process
full-batch BatchNorm (torch autograd, float64)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as TF


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from segmentron_amd import parallel
        torch.manual_seed(0)  # same full batch on every rank; each takes its shard
        N, C, H, W = 4, 24, 5, 7
        x = (torch.randn(N, C, H, W, dtype=torch.float64) * 1.7 + 0.4).requires_grad_()
        gamma = (torch.rand(C, dtype=torch.float64) + 0.5).requires_grad_()
        beta = torch.randn(C, dtype=torch.float64).requires_grad_()
        g = torch.randn(N, C, H, W, dtype=torch.float64)
        eps = 1e-3
        y = torch.relu(TF.batch_norm(x, None, None, gamma, beta, True, 0.1, eps))
        y.backward(g)

        sync = torch.nn.SyncBatchNorm(C).train()
        group = parallel.sync_group(sync)
        assert group is not None
        assert parallel.sync_group(torch.nn.BatchNorm2d(C).train()) is None
        assert parallel.sync_group(torch.nn.SyncBatchNorm(C).eval()) is None

        # UNEQUAL shards (1 and 3 images): the element count travels with the sums
        lo, hi = (0, 1) if rank == 0 else (1, N)
        xs = x.detach()[lo:hi]
        gs = g[lo:hi]
        # ---- forward exchange
        sums = torch.cat([xs.sum((0, 2, 3)), (xs * xs).sum((0, 2, 3))])
        sums, cnt = parallel.allreduce_forward_sums(sums, xs.numel() // C, group)
        assert cnt == N * H * W
        mean = sums[:C] / cnt
        var = sums[C:] / cnt - mean * mean
        invstd = 1.0 / torch.sqrt(var + eps)
        xd = x.detach()
        assert torch.allclose(mean, xd.mean((0, 2, 3)), rtol=1e-12, atol=1e-12)
        assert torch.allclose(var, xd.var((0, 2, 3), unbiased=False), rtol=1e-10, atol=1e-12)
        scale = gamma.detach() * invstd
        shift = beta.detach() - mean * scale
        # ---- backward exchange (formulas of seg_bn_bwd_finalize / seg_bn_bwd_apply)
        ylin = xs * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
        gp = gs * (ylin > 0)
        bs = torch.cat([gp.sum((0, 2, 3)), (gp * xs).sum((0, 2, 3))])
        bs = parallel.allreduce_backward_sums(bs, group)
        sg, sgx = bs[:C], bs[C:]
        dgamma = (sgx - mean * sg) * invstd
        dbeta = sg
        c1 = scale * (dgamma / cnt) * invstd
        c0 = scale * (sg / cnt) - c1 * mean
        dx = scale.view(1, -1, 1, 1) * gp - c0.view(1, -1, 1, 1) - c1.view(1, -1, 1, 1) * xs
        assert torch.allclose(dx, x.grad[lo:hi], rtol=1e-9, atol=1e-11)
        # ---- parameter gradients as DistributedDataParallel will see them
        dgl, dbl = parallel.local_param_grads(dgamma, dbeta, group)
        avg = torch.stack([dgl, dbl])
        dist.all_reduce(avg)
        avg /= world  # DDP averages
        assert torch.allclose(avg[0], gamma.grad / world, rtol=1e-9, atol=1e-11)
        assert torch.allclose(avg[1], beta.grad / world, rtol=1e-9, atol=1e-11)
        # ---- SMALL SyncBatchNorm (ASPP image pooling: a few samples per rank with a mean far
        # from zero): per-rank two-pass moments merged as float64 sums (what
        # seg_bn_moments_small / seg_bn_finalize_small_sync compute) reproduce the full-batch
        # two-pass variance where fp32 (sum x, sum x^2) rows lose it entirely
        base = torch.full((C,), 37.5, dtype=torch.float64)
        small = (base.view(1, C) + torch.randn(N, C, dtype=torch.float64) * 1e-3).float().double()
        part = small[lo:hi]
        n_r = float(part.shape[0])
        m_r = part.mean(0)
        m2_r = ((part - m_r) ** 2).sum(0)
        buf = torch.cat([n_r * m_r, m2_r + n_r * m_r * m_r, torch.tensor([n_r], dtype=torch.float64)])
        msum, mcnt = parallel.allreduce_moments(buf, group)
        assert float(mcnt) == N
        gmean = msum[:C] / mcnt
        gvar = msum[C:] / mcnt - gmean * gmean
        want_var = small.var(0, unbiased=False)
        assert torch.allclose(gmean, small.mean(0), rtol=1e-14, atol=0)
        assert (gvar - want_var).abs().max() < 1e-3 * want_var.min() + 1e-12, \
            (gvar - want_var).abs().max()
        f32 = small.float()
        bad_var = (f32 * f32).sum(0) / N - (f32.sum(0) / N) ** 2  # the fp32 single-pass form
        assert (bad_var.double() - want_var).abs().max() > 10 * want_var.max()
        # ---- the reference's own NaiveSyncBatchNorm (modules/batch_norm.py:150-183): an
        # nn.BatchNorm2d that syncs when training on > 1 rank, biased running_var, no counter
        from segmentron_amd.modules.batch_norm import NaiveSyncBatchNorm
        nb = NaiveSyncBatchNorm(C, eps=eps, momentum=0.1).double().train()
        assert isinstance(nb, torch.nn.BatchNorm2d) and not isinstance(nb, torch.nn.SyncBatchNorm)
        assert parallel.sync_group(nb) is dist.group.WORLD
        assert parallel.sync_group(NaiveSyncBatchNorm(C).eval()) is None
        parallel.naive_running_update(nb, mean, invstd)
        assert torch.allclose(nb.running_mean, 0.1 * mean, rtol=1e-12, atol=1e-14)
        assert torch.allclose(nb.running_var, 1 + 0.1 * (var - 1), rtol=1e-9, atol=1e-12)
        assert int(nb.num_batches_tracked) == 0
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


def test_syncbn_exchange_protocol_world_size_2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: 1, 1: 1}


def test_stats_exchange_routing_and_mailbox_lookup():
    """Host logic of segmentron_amd/xgmi.py + parallel.mailbox() (no GPU): float64 / float32 sums
    that fit a slot go to the mailbox, everything else — other ops, other dtypes, oversized
    vectors, grouped gradient calls — to the wrapped communicator; `parallel.mailbox()` is what
    switches functional.finish_bn to the in-kernel exchange."""
    from segmentron_amd import parallel, xgmi
    calls = []

    class Box:
        rank, world, slot_bytes = 0, 2, 64

        def fits(self, t):
            return t.dtype in (torch.float64, torch.float32) and \
                0 < t.numel() * t.element_size() <= self.slot_bytes

        def all_reduce(self, t):
            calls.append(("box", t.dtype, t.numel()))
            return t

    class Comm:
        def all_reduce(self, t, op="sum"):
            calls.append(("comm", op, t.dtype, t.numel()))
            return t

        def all_reduce_many(self, ts, op="sum"):
            calls.append(("comm-many", op, len(ts)))

    ex = xgmi.StatsExchange(Box(), Comm())
    ex.all_reduce(torch.zeros(8, dtype=torch.float64))
    ex.all_reduce(torch.zeros(16, dtype=torch.float32))
    ex.all_reduce(torch.zeros(9, dtype=torch.float64))           # 72 bytes > slot
    ex.all_reduce(torch.zeros(4, dtype=torch.float64), "avg")    # not a sum
    ex.all_reduce(torch.zeros(4, dtype=torch.bfloat16))
    ex.all_reduce_many([torch.zeros(3), torch.zeros(5)], "avg")
    assert calls == [("box", torch.float64, 8), ("box", torch.float32, 16),
                     ("comm", "sum", torch.float64, 9), ("comm", "avg", torch.float64, 4),
                     ("comm", "sum", torch.bfloat16, 4), ("comm-many", "avg", 2)]
    assert parallel.mailbox() is None
    prev = parallel.use_native_rccl(ex)
    try:
        assert parallel.mailbox() is ex.mailbox
        parallel.use_native_rccl(Comm())
        assert parallel.mailbox() is None  # a plain communicator: three-launch exchange path
    finally:
        parallel.use_native_rccl(prev)


def _subgroup_worker(rank, world, port, ret):
    """with a native communicator installed, a SyncBatchNorm over a process SUBGROUP
    must reduce over that subgroup (torch.distributed), not over the native world communicator
    or its mailbox; world-wide BatchNorms keep the native path."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from segmentron_amd import parallel
        groups = [dist.new_group([r]) for r in range(world)]  # collective: every rank, same order
        calls = []

        class Box:
            rank, world = 0, 2

        class Native:
            world, mailbox = 2, Box()

            def all_reduce(self, t, op="sum"):
                calls.append(t.numel())
                dist.all_reduce(t)
                return t

        prev = parallel.use_native_rccl(Native())
        try:
            mine = groups[rank]
            sync = torch.nn.SyncBatchNorm(4, process_group=mine).train()
            assert parallel.sync_group(sync) is mine
            assert parallel.mailbox(mine) is None and parallel.mailbox(dist.group.WORLD) is Native.mailbox
            assert parallel.mailbox(None) is Native.mailbox
            t = torch.full((6,), float(rank + 1), dtype=torch.float64)
            sums, cnt = parallel.allreduce_forward_sums(t.clone(), 5, mine)
            assert calls == [] and torch.equal(sums, t) and float(cnt) == 5.0  # own rank only
            b = parallel.allreduce_backward_sums(t.clone(), mine)
            assert calls == [] and torch.equal(b, t)
            assert parallel.grad_scale(mine) == 1.0
            sums, cnt = parallel.allreduce_forward_sums(t.clone(), 5, dist.group.WORLD)
            assert calls == [7] and torch.equal(sums, torch.full((6,), 3.0, dtype=torch.float64))
            assert float(cnt) == 10.0 and parallel.grad_scale(dist.group.WORLD) == 0.5
        finally:
            parallel.use_native_rccl(prev)
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


def test_subgroup_syncbn_does_not_use_the_world_communicator():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_subgroup_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: 1, 1: 1}


def test_overlapped_averager_rejects_a_second_backward_without_finish():
    """gradient accumulation through the overlapped averager must not silently mix
    averaged and local gradients."""
    import pytest
    from segmentron_amd import parallel

    class FakeStream:
        def wait_stream(self, s):
            pass

    calls = []

    class Comm:
        def all_reduce_many(self, ts, op="sum"):
            calls.append(len(ts))

    av = parallel.OverlappedGradientAverager.__new__(parallel.OverlappedGradientAverager)
    ps = [torch.nn.Parameter(torch.zeros(3)) for _ in range(3)]
    av.comm, av.side = Comm(), FakeStream()
    av.buckets = [[ps[2], ps[1]], [ps[0]]]
    av._bucket_of = {id(ps[2]): 0, id(ps[1]): 0, id(ps[0]): 1}
    av._left, av._launched, av._paused, av._hooks = [2, 1], [False, False], False, []
    launched = []
    av._launch = lambda bi: (launched.append(bi), av._launched.__setitem__(bi, True))
    av._ready(ps[2])
    av._ready(ps[1])
    assert launched == [0]
    with pytest.raises(RuntimeError):
        av._ready(ps[2])  # second backward before finish()
    av._left, av._launched = [2, 1], [False, False]
    with av.no_sync():
        av._ready(ps[2])
        av._ready(ps[1])
    assert launched == [0] and av._left == [2, 1]  # accumulation pass: nothing launched
    av._ready(ps[2])
    av._ready(ps[1])
    assert launched == [0, 0]
