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


def is_naive_sync(bn):
    """The reference's own NaiveSyncBatchNorm (an nn.BatchNorm2d subclass)."""
    return type(bn).__name__ == "NaiveSyncBatchNorm" and isinstance(bn, nn.BatchNorm2d)


def sync_group(bn):
    """Process group to synchronise over, or None (plain BN / eval / single process)."""
    if not (bn.training and dist.is_available() and dist.is_initialized()
            and dist.get_world_size() > 1):
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


def allreduce_forward_sums(sums, local_count, group):
    """sums: float64 [2C] local (sum x, sum x^2) -> (global sums, global count).
    The local element count rides in the same message ([2C+1]), so ranks with different
    N*H*W (uneven last batch, variable-size inputs) still get the exact global statistics —
    torch's SyncBatchNorm all-gathers per-rank counts for the same reason
    (torch/nn/modules/_functions.py:49-74)."""
    import torch
    n = sums.numel()
    buf = torch.empty(n + 1, dtype=sums.dtype, device=sums.device)
    buf[:n] = sums
    buf[n] = float(local_count)
    dist.all_reduce(buf, group=group)
    return buf[:n], buf[n:]


def allreduce_backward_sums(sums, group):
    """sums: [2C] local (sum g', sum g'*x) or (ds, dt) -> global, in place."""
    dist.all_reduce(sums, group=group)
    return sums


def local_param_grads(dgamma, dbeta, group):
    """Every rank computed dgamma/dbeta from GLOBAL sums; DistributedDataParallel will average
    parameter gradients over ranks, and torch's SyncBatchNorm returns LOCAL sums there — divide
    so that DDP's mean reproduces the reference value (global / world)."""
    ws = dist.get_world_size(group)
    return dgamma / ws, dbeta / ws
