"""Norm-layer selection — interface of segmentron/modules/batch_norm.py:111-132 (`get_norm`).

All BatchNorm variants are plain parameter containers here (nn.BatchNorm2d / nn.SyncBatchNorm
subclasses keep the reference's state_dict keys and stay visible to
`nn.SyncBatchNorm.convert_sync_batchnorm` and `_set_batch_norm_attr`); the
arithmetic is done by the HIP kernels through segmentron_amd.functional.finish_bn, which reads
eps / momentum / training at call time and all-reduces the statistics over RCCL when the module
is a SyncBatchNorm."""
import torch.nn as nn


class NaiveSyncBatchNorm(nn.BatchNorm2d):
    """'SyncBN' of the reference (segmentron/modules/batch_norm.py:150-183).  Like there it IS an
    nn.BatchNorm2d (so `_set_batch_norm_attr`, `convert_sync_batchnorm` and state_dicts treat it
    as one) and behaves as plain BatchNorm in eval mode / with one process.  Training with
    world_size > 1: cross-rank batch statistics (equal weight per rank), `running_var` updated
    with the BIASED variance and `num_batches_tracked` left alone (`:172-176` never calls
    `super().forward`).  Served by the same fused statistics all-reduce as nn.SyncBatchNorm
    (segmentron_amd.functional.finish_bn / parallel.naive_running_update)."""


def get_norm(norm):
    support = ["BN", "SyncBN", "nnSyncBN"]
    unsupported = ["FrozenBN", "GN"]
    if isinstance(norm, str):
        if norm in unsupported:
            raise NotImplementedError(
                "BN_TYPE %r is outside the MI355X hot path (only %s are served by HIP kernels)"
                % (norm, support))
        assert norm in support, "Unknown norm type {}, support norm types are {}".format(
            norm, support + unsupported)
        return {"BN": nn.BatchNorm2d, "SyncBN": NaiveSyncBatchNorm,
                "nnSyncBN": nn.SyncBatchNorm}[norm]
    return norm
