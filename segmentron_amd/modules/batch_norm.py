"""Norm-layer selection — interface of segmentron/modules/batch_norm.py:111-132 (`get_norm`).

All BatchNorm variants are plain parameter containers here (nn.BatchNorm2d / nn.SyncBatchNorm
subclasses keep the reference's state_dict keys and stay visible to
`nn.SyncBatchNorm.convert_sync_batchnorm` and `_set_batch_norm_attr`); the
arithmetic is done by the HIP kernels through segmentron_amd.functional.finish_bn, which reads
eps / momentum / training at call time and all-reduces the statistics over RCCL when the module
is a SyncBatchNorm."""
import logging

import torch
import torch.nn as nn


class NaiveSyncBatchNorm(nn.BatchNorm2d):
    """'SyncBN' of the reference (segmentron/modules/batch_norm.py:150-183).  Like there it IS an
    nn.BatchNorm2d (so `_set_batch_norm_attr`, `convert_sync_batchnorm` and state_dicts treat it
    as one) and behaves as plain BatchNorm in eval mode / with one process.  Training with
    world_size > 1: cross-rank batch statistics (equal weight per rank), `running_var` updated
    with the BIASED variance and `num_batches_tracked` left alone (`:172-176` never calls
    `super().forward`).  Served by the same fused statistics all-reduce as nn.SyncBatchNorm
    (segmentron_amd.functional.finish_bn / parallel.naive_running_update)."""


class FrozenBatchNorm2d(nn.Module):
    """'FrozenBN' of the reference (segmentron/modules/batch_norm.py:10-104): BatchNorm whose
    statistics AND affine parameters are fixed — four non-trainable BUFFERS (weight, bias,
    running_mean, running_var; no num_batches_tracked), `y = x * scale + shift` in every mode.
    Same state_dict keys, the same `_version = 3` loading rules (version < 2: missing running
    statistics are filled in, version < 3: `running_var -= eps`) and `convert_frozen_batchnorm`.
    On the HIP path it is a constant per-channel affine pending on the producing convolution
    (`functional.finish_bn` -> seg_bn_eval_affine; `frozen` makes every consumer treat it as an
    evaluation-mode BatchNorm, so it folds into a following 1x1 convolution like one)."""
    _version = 3
    frozen = True

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)

    # what functional.finish_bn reads of an nn.BatchNorm2d
    momentum, track_running_stats, num_batches_tracked, affine = 0.0, False, None, True

    def forward(self, x):
        raise RuntimeError("FrozenBatchNorm2d is applied by the HIP kernels through "
                           "segmentron_amd.functional (deferred affine), not called directly")

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys,
                              unexpected_keys, error_msgs):
        version = local_metadata.get("version", None)
        if version is None or version < 2:
            if prefix + "running_mean" not in state_dict:
                state_dict[prefix + "running_mean"] = torch.zeros_like(self.running_mean)
            if prefix + "running_var" not in state_dict:
                state_dict[prefix + "running_var"] = torch.ones_like(self.running_var)
        if version is not None and version < 3:
            logging.info("FrozenBatchNorm {} is upgraded to version 3.".format(prefix.rstrip(".")))
            state_dict[prefix + "running_var"] -= self.eps
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys,
                                      unexpected_keys, error_msgs)

    def __repr__(self):
        return "FrozenBatchNorm2d(num_features={}, eps={})".format(self.num_features, self.eps)

    @classmethod
    def convert_frozen_batchnorm(cls, module):
        """BatchNorm2d / SyncBatchNorm children -> FrozenBatchNorm2d (batch_norm.py:72-104;
        `running_var + eps` is stored, as there)."""
        res = module
        if isinstance(module, (nn.BatchNorm2d, nn.SyncBatchNorm)):
            res = cls(module.num_features)
            if module.affine:
                res.weight.data = module.weight.data.clone().detach()
                res.bias.data = module.bias.data.clone().detach()
            res.running_mean.data = module.running_mean.data
            res.running_var.data = module.running_var.data + module.eps
        else:
            for name, child in module.named_children():
                new_child = cls.convert_frozen_batchnorm(child)
                if new_child is not child:
                    res.add_module(name, new_child)
        return res


def groupNorm(num_channels, eps=1e-5, momentum=0.1, affine=True):
    """The reference's `GN` factory (batch_norm.py:105-108): nn.GroupNorm(min(32, C), C).
    Served by csrc/groupnorm.hip through functional.conv_bn / dwconv_bn (materialised: group
    statistics are per sample)."""
    return nn.GroupNorm(min(32, num_channels), num_channels, eps=eps, affine=affine)


def get_norm(norm):
    support = ["BN", "SyncBN", "nnSyncBN", "FrozenBN", "GN"]
    if isinstance(norm, str):
        assert norm in support, "Unknown norm type {}, support norm types are {}".format(
            norm, support)
        return {"BN": nn.BatchNorm2d, "SyncBN": NaiveSyncBatchNorm,
                "nnSyncBN": nn.SyncBatchNorm, "FrozenBN": FrozenBatchNorm2d,
                "GN": groupNorm}[norm]
    return norm
