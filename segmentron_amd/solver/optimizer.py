"""`get_optimizer(model)` of segmentron/solver/optimizer.py:14-66 with the SGD step on a
multi-tensor HIP kernel.

Parameter groups, the encoder/decoder LR split (DECODER_LR_FACTOR) and the BatchNorm eps /
momentum attributes are set exactly as the reference does (optimizer.py:14-41).  For
SOLVER.OPTIMIZER == 'sgd' the returned object is `FusedSGD`: a `torch.optim.Optimizer` with
torch.optim.SGD's constructor, `param_groups` and `state_dict()` layout (`momentum_buffer`), whose
`step()` is ONE launch per 48 tensors (csrc/optim.hip) instead of torch's per-tensor foreach
chain.  The other optimizers the reference offers (adam / adadelta / rmsprop) are host plumbing
around torch's own implementations and are passed through unchanged.
"""
import logging

import torch
import torch.nn as nn
from torch import optim

from .. import hip_ops as K
from ..config import cfg

__all__ = ["FusedSGD", "get_optimizer"]


class FusedSGD(optim.Optimizer):
    """torch.optim.SGD(params, lr, momentum, weight_decay) — dampening 0, no Nesterov, which is
    all the reference uses — on the HIP multi-tensor kernel.  The per-group learning rates and
    weight decays live in a small device tensor that `step()` refreshes from `param_groups`
    whenever they changed, so an LR scheduler works unmodified; inside a captured HIP graph the
    kernel keeps reading that tensor — call `sync_hyperparameters()` after `scheduler.step()`
    between replays."""

    def __init__(self, params, lr=1e-3, momentum=0.0, weight_decay=0.0, dampening=0.0,
                 nesterov=False):
        if dampening != 0.0 or nesterov:
            raise ValueError("FusedSGD implements dampening=0, nesterov=False "
                             "(segmentron/solver/optimizer.py:48-49)")
        if lr < 0.0 or momentum < 0.0 or weight_decay < 0.0:
            raise ValueError("invalid lr / momentum / weight_decay")
        defaults = dict(lr=lr, momentum=momentum, weight_decay=weight_decay, dampening=0.0,
                        nesterov=False)
        super().__init__(params, defaults)
        if len({g["momentum"] for g in self.param_groups}) > 1:
            raise ValueError("FusedSGD: one momentum for all parameter groups")
        self._hyper_dev, self._hyper_host = None, None
        self._plans = {}

    # -- device copy of (lr, weight_decay) per group
    def sync_hyperparameters(self):
        host = [float(g["lr"]) for g in self.param_groups] + \
               [float(g["weight_decay"]) for g in self.param_groups]
        if self._hyper_dev is None:
            dev = None
            for g in self.param_groups:
                for p in g["params"]:
                    dev = p.device
                    break
                if dev is not None:
                    break
            if dev is None or dev.type != "cuda":
                raise RuntimeError("FusedSGD needs parameters on a HIP device (no CPU fallback)")
            self._hyper_dev = torch.empty(len(host), dtype=torch.float32, device=dev)
        if host != self._hyper_host:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("FusedSGD: lr / weight_decay changed during graph capture")
            self._hyper_dev.copy_(torch.tensor(host, dtype=torch.float32))
            self._hyper_host = host
        ng = len(self.param_groups)
        return self._hyper_dev[:ng], self._hyper_dev[ng:]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lr_dev, wd_dev = self.sync_hyperparameters()
        momentum = self.param_groups[0]["momentum"]
        # parameters whose momentum buffer does not exist yet take the `first` form of the
        # update (buf = d), exactly like torch (sgd.py: buf = clone(d_p)); both kinds can occur
        # in one step when a parameter received its first gradient late
        fresh, old = ([], [], [], []), ([], [], [], [])
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                st = self.state[p]
                buf = st.get("momentum_buffer") if momentum != 0.0 else st.get("_scratch")
                dst = old if (buf is not None and momentum != 0.0) else fresh
                if buf is None:
                    if g.is_sparse:
                        raise RuntimeError("FusedSGD does not support sparse gradients")
                    buf = torch.empty_like(p, memory_format=torch.contiguous_format)
                    st["momentum_buffer" if momentum != 0.0 else "_scratch"] = buf
                if not g.is_contiguous():
                    g = g.contiguous()
                dst[0].append(p)
                dst[1].append(g)
                dst[2].append(buf)
                dst[3].append(gi)
        for first, (ps, gs, bs, gi) in ((True, fresh), (False, old)):
            if not ps:
                continue
            # the pointer tables of parameters / buffers are rebuilt only when the set changes
            key = (first, len(ps), ps[0].data_ptr(), ps[-1].data_ptr(), bs[0].data_ptr(),
                   bs[-1].data_ptr())
            plan = self._plans.get(first)
            if plan is None or plan[0] != key or any(a is not b for a, b in zip(plan[1], ps)):
                for p, g, b in zip(ps, gs, bs):
                    K.sgd_check(p, g, b)
                plan = (key, list(ps), K.sgd_plan(ps, bs, gi))
                self._plans[first] = plan
            else:
                g0 = gs[0]
                if not (g0.is_cuda and g0.dtype == torch.float32):
                    raise RuntimeError("segmentron_amd SGD: gradients must be fp32 HIP tensors")
            K.sgd_multi_tensor(plan[2], gs, lr_dev, wd_dev, momentum, first)
            # the kernel writes the parameters through raw pointers: tell autograd (and
            # functional.cached_pack, which keys the packed weight copies on `_version`)
            torch.autograd.graph.increment_version(ps)
        return loss

    def state_dict(self):
        sd = super().state_dict()
        # the momentum-free scratch is not optimizer state; the per-parameter dicts returned by
        # Optimizer.state_dict() are the LIVE ones, so filter into copies
        sd["state"] = {k: {n: v for n, v in st.items() if n != "_scratch"}
                       for k, st in sd["state"].items()}
        return sd


def _set_batch_norm_attr(named_modules, attr, value):
    for _, m in named_modules:
        if isinstance(m, (nn.BatchNorm2d, nn.SyncBatchNorm)):
            setattr(m, attr, value)


def _get_parameters(model):
    """optimizer.py:14-41: encoder at SOLVER.LR, every decoder module at LR * DECODER_LR_FACTOR,
    custom BatchNorm eps for encoder / decoder, custom BatchNorm momentum."""
    params_list = []
    if hasattr(model, "encoder") and model.encoder is not None and hasattr(model, "decoder"):
        params_list.append({"params": model.encoder.parameters(), "lr": cfg.SOLVER.LR})
        if cfg.MODEL.BN_EPS_FOR_ENCODER:
            logging.info("Set bn custom eps for bn in encoder: {}".format(cfg.MODEL.BN_EPS_FOR_ENCODER))
            _set_batch_norm_attr(model.encoder.named_modules(), "eps", cfg.MODEL.BN_EPS_FOR_ENCODER)
        for module in model.decoder:
            params_list.append({"params": getattr(model, module).parameters(),
                                "lr": cfg.SOLVER.LR * cfg.SOLVER.DECODER_LR_FACTOR})
        if cfg.MODEL.BN_EPS_FOR_DECODER:
            logging.info("Set bn custom eps for bn in decoder: {}".format(cfg.MODEL.BN_EPS_FOR_DECODER))
            for module in model.decoder:
                _set_batch_norm_attr(getattr(model, module).named_modules(), "eps",
                                     cfg.MODEL.BN_EPS_FOR_DECODER)
    else:
        logging.info("Model do not have encoder or decoder, params list was from model.parameters(), "
                     "and arguments BN_EPS_FOR_ENCODER, BN_EPS_FOR_DECODER, DECODER_LR_FACTOR not used!")
        params_list = model.parameters()
    if cfg.MODEL.BN_MOMENTUM and cfg.MODEL.BN_TYPE in ["BN"]:
        logging.info("Set bn custom momentum: {}".format(cfg.MODEL.BN_MOMENTUM))
        _set_batch_norm_attr(model.named_modules(), "momentum", cfg.MODEL.BN_MOMENTUM)
    elif cfg.MODEL.BN_MOMENTUM and cfg.MODEL.BN_TYPE not in ["BN"]:
        logging.info("Batch norm type is {}, custom bn momentum is not effective!".format(cfg.MODEL.BN_TYPE))
    return params_list


def get_optimizer(model):
    parameters = _get_parameters(model)
    opt_lower = cfg.SOLVER.OPTIMIZER.lower()
    if opt_lower == "sgd":
        return FusedSGD(parameters, lr=cfg.SOLVER.LR, momentum=cfg.SOLVER.MOMENTUM,
                        weight_decay=cfg.SOLVER.WEIGHT_DECAY)
    if opt_lower == "adam":
        return optim.Adam(parameters, lr=cfg.SOLVER.LR, eps=cfg.SOLVER.EPSILON,
                          weight_decay=cfg.SOLVER.WEIGHT_DECAY)
    if opt_lower == "adadelta":
        return optim.Adadelta(parameters, lr=cfg.SOLVER.LR, eps=cfg.SOLVER.EPSILON,
                              weight_decay=cfg.SOLVER.WEIGHT_DECAY)
    if opt_lower == "rmsprop":
        return optim.RMSprop(parameters, lr=cfg.SOLVER.LR, alpha=0.9, eps=cfg.SOLVER.EPSILON,
                             momentum=cfg.SOLVER.MOMENTUM, weight_decay=cfg.SOLVER.WEIGHT_DECAY)
    raise ValueError("Expected optimizer method in [sgd, adam, adadelta, rmsprop], but received "
                     "{}".format(opt_lower))
