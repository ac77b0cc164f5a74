"""LR schedules of segmentron/solver/lr_scheduler.py (poly :13-43, multi-step :46-80, cosine
:83-121, selector :150-168) as ONE closed-form scheduler: lr(t) = target + (base - target) *
warmup(t) * decay(t).  Host arithmetic only — with `FusedSGD` the value reaches the kernel through
its device-resident hyper-parameter tensor (`optimizer.sync_hyperparameters()`), so a captured
step graph follows the schedule."""
import math
from bisect import bisect_right

import torch

from ..config import cfg

__all__ = ["WarmupLR", "WarmupPolyLR", "WarmupMultiStepLR", "WarmupCosineLR", "get_scheduler"]


def _warmup_factor(method, it, warmup_iters, factor):
    """1 past the warm-up; `factor` (constant) or a ramp from `factor` to 1 (linear) inside."""
    if it >= warmup_iters:
        return 1.0
    if method == "constant":
        return factor
    if method == "linear":
        alpha = float(it) / warmup_iters
        return factor * (1 - alpha) + alpha
    raise ValueError("Only 'constant' or 'linear' warmup_method accepted got {}".format(method))


class WarmupLR(torch.optim.lr_scheduler.LRScheduler):
    """decay = 'poly' | 'cosine' | 'step'.  The poly law replaces the decay by the warm-up ramp
    during warm-up and counts its horizon from the END of the warm-up (lr_scheduler.py:31-43);
    the other two multiply ramp and decay and count from iteration 0 (:68-75, :100-116)."""

    def __init__(self, optimizer, decay, max_iters=0, power=0.9, milestones=(), gamma=0.1,
                 target_lr=0.0, warmup_factor=1.0 / 3, warmup_iters=0, warmup_method="linear",
                 last_epoch=-1):
        if warmup_method not in ("constant", "linear"):
            raise ValueError("Only 'constant' or 'linear' warmup_method accepted "
                             "got {}".format(warmup_method))
        if decay not in ("poly", "cosine", "step"):
            raise ValueError("not support lr scheduler method!")
        if list(milestones) != sorted(milestones):
            raise ValueError("Milestones should be a list of increasing integers. Got {}".format(milestones))
        self.decay, self.max_iters, self.power = decay, max_iters, power
        self.milestones, self.gamma, self.target_lr = list(milestones), gamma, target_lr
        self.warmup_factor, self.warmup_iters = warmup_factor, warmup_iters
        self.warmup_method = warmup_method
        super().__init__(optimizer, last_epoch)

    def _scale(self, t):
        w = _warmup_factor(self.warmup_method, t, self.warmup_iters, self.warmup_factor)
        if self.decay == "poly":
            if t < self.warmup_iters:
                return w
            return pow(1 - (t - self.warmup_iters) / (self.max_iters - self.warmup_iters),
                       self.power)
        if self.decay == "cosine":
            return w * 0.5 * (1.0 + math.cos(math.pi * t / self.max_iters))
        return w * self.gamma ** bisect_right(self.milestones, t)

    def get_lr(self):
        s = self._scale(self.last_epoch)
        return [self.target_lr + (b - self.target_lr) * s for b in self.base_lrs]

    def step(self, epoch=None):
        super().step(epoch) if epoch is not None else super().step()
        # FusedSGD: push the new values to the device copy its kernel reads (if it exists yet —
        # otherwise the first step() creates it from param_groups)
        if getattr(self.optimizer, "_hyper_dev", None) is not None \
                and not torch.cuda.is_current_stream_capturing():
            self.optimizer.sync_hyperparameters()


class WarmupPolyLR(WarmupLR):
    def __init__(self, optimizer, target_lr=0, max_iters=0, power=0.9, warmup_factor=1.0 / 3,
                 warmup_iters=500, warmup_method="linear", last_epoch=-1):
        super().__init__(optimizer, "poly", max_iters=max_iters, power=power, target_lr=target_lr,
                         warmup_factor=warmup_factor, warmup_iters=warmup_iters,
                         warmup_method=warmup_method, last_epoch=last_epoch)


class WarmupMultiStepLR(WarmupLR):
    def __init__(self, optimizer, milestones, gamma=0.1, warmup_factor=0.001, warmup_iters=1000,
                 warmup_method="linear", last_epoch=-1):
        super().__init__(optimizer, "step", milestones=milestones, gamma=gamma,
                         warmup_factor=warmup_factor, warmup_iters=warmup_iters,
                         warmup_method=warmup_method, last_epoch=last_epoch)


class WarmupCosineLR(WarmupLR):
    def __init__(self, optimizer, max_iters, warmup_factor=0.001, warmup_iters=1000,
                 warmup_method="linear", last_epoch=-1):
        super().__init__(optimizer, "cosine", max_iters=max_iters, warmup_factor=warmup_factor,
                         warmup_iters=warmup_iters, warmup_method=warmup_method,
                         last_epoch=last_epoch)


def get_scheduler(optimizer, max_iters, iters_per_epoch):
    mode = cfg.SOLVER.LR_SCHEDULER.lower()
    warm = iters_per_epoch * cfg.SOLVER.WARMUP.EPOCHS
    common = dict(warmup_factor=cfg.SOLVER.WARMUP.FACTOR, warmup_iters=warm,
                  warmup_method=cfg.SOLVER.WARMUP.METHOD)
    if mode == "poly":
        return WarmupPolyLR(optimizer, max_iters=max_iters, power=cfg.SOLVER.POLY.POWER, **common)
    if mode == "cosine":
        return WarmupCosineLR(optimizer, max_iters=max_iters, **common)
    if mode == "step":
        return WarmupMultiStepLR(optimizer, milestones=[x * iters_per_epoch
                                                        for x in cfg.SOLVER.STEP.DECAY_EPOCH],
                                 gamma=cfg.SOLVER.STEP.GAMMA, **common)
    raise ValueError("not support lr scheduler method!")
