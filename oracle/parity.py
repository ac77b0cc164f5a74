"""Parity of one FULL-SIZE train step of the benched configuration (TEST INFRASTRUCTURE ONLY).

BASELINE.json configs[2] — DeepLabv3+ xception65, train, batch 2 @1025x2049 — is the workload
`bench.py` times; this module compares ONE such step of the HIP path (fp32 kernels and the bf16
throughput path) with the CPU oracle's step on the same state and input
(/root/reference/tools/train.py:135-146 -> models/deeplabv3_plus.py:33-46 -> solver/loss.py:16-46).
It is shared by `bench.py`'s `cpu_baseline` leg (which runs the oracle step anyway and used to
throw its result away — VERDICT r04 Missing #1) and by
tests/test_parity_conditioned.py::test_c3_train_full_size_1025x2049_matches_oracle.

State: `oracle.synth(conditioned=True)`, the well-conditioned state of the fixed-bar parity tests
(oracle/synth.py header: on the default random state a 70-layer ReLU+BatchNorm chain is a chaotic
map and no fixed bar means anything), dropout off on both sides.

Never imported by the product package (`segmentron_amd/`): only tests/ and bench.py's
cpu_baseline leg call it, and only as the checker.
"""
import time

import torch

from . import synth, torch_ref

SAMPLE = 16  # logits are compared on a [::16, ::16] pixel grid (2 x 19 x 65 x 129 values)


def conditioned_state(model_state_dict, seed=0):
    return synth.synth_like(model_state_dict, seed=seed, conditioned=True)


def inputs(batch, h, w, seed=0):
    return synth.synth_images(batch, h, w, seed=seed), synth.synth_targets(batch, h, w, seed=seed)


AUX_WEIGHT = 0.4  # cfg.SOLVER.AUX_WEIGHT default (segmentron/config/settings.py)


def oracle_step(sd, x, y, dtype=torch.float32, oracle_fn="deeplabv3_plus_xception65", **net_kw):
    """One oracle train step (forward + MixSoftmaxCrossEntropyLoss + backward) ->
    dict(loss, logits [::SAMPLE] sample, argmax sample, grads {name: tensor}, seconds).
    net_kw: OracleNet arguments (output_stride / aux for PSPNet, eps_encoder=None for ResNets)."""
    s = {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    s = torch_ref.clone_state(s, requires_grad=True)
    kw = dict(eps_encoder=1e-3, drop_p=0.0)
    kw.update(net_kw)
    net = torch_ref.OracleNet(s, training=True, **kw)
    t0 = time.perf_counter()
    outs = getattr(net, oracle_fn)(x.to(dtype))
    loss = torch_ref.mix_softmax_ce(outs, y, aux_weight=AUX_WEIGHT)
    loss.backward()
    secs = time.perf_counter() - t0
    lo = outs[0].detach()[..., ::SAMPLE, ::SAMPLE].float().clone()
    return dict(loss=float(loss.item()), logits=lo, grads={k: v.grad.detach() for k, v in s.items()
                                                           if v.grad is not None}, seconds=secs)


def hip_step(dtype, sd, x, y, dev="cuda", eps_encoder=1e-3):
    """The same step on a FRESH HIP model of the current cfg in `dtype` ('fp32' | 'bf16'), eager
    launches -> dict(loss, logits sample, grads on the CPU).  Auxiliary heads (PSPNet with
    SOLVER.AUX) enter the loss with the reference's weight (solver/loss.py:16-46)."""
    import segmentron_amd
    from segmentron_amd import functional as SF
    prev = segmentron_amd.compute_dtype()
    SF.clear_weight_cache()
    segmentron_amd.set_compute_dtype(dtype)
    try:
        model = segmentron_amd.get_segmentation_model()
        model.load_state_dict(sd, strict=True)
        if eps_encoder is not None:
            for _, m in model.encoder.named_modules():  # solver/optimizer.py:18-20
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.eps = eps_encoder
        model = model.to(dev).train()
        for m in model.modules():
            if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)):
                m.p = 0.0
        outs = model(x.to(dev))
        yd = y.to(dev)
        loss = torch.nn.functional.cross_entropy(outs[0], yd, ignore_index=-1)
        for aux in outs[1:]:
            loss = loss + AUX_WEIGHT * torch.nn.functional.cross_entropy(aux, yd, ignore_index=-1)
        loss.backward()
        torch.cuda.synchronize()
        res = dict(loss=float(loss.item()),
                   logits=outs[0].detach()[..., ::SAMPLE, ::SAMPLE].float().cpu(),
                   grads={k: p.grad.detach().float().cpu() for k, p in model.named_parameters()
                          if p.grad is not None})
        del model, outs, loss
        return res
    finally:
        SF.clear_weight_cache()
        segmentron_amd.set_compute_dtype(prev)
        torch.cuda.empty_cache()


def compare(got, ref):
    """-> dict of the figures the bars are set on (all against the oracle step `ref`)."""
    a, b = got["logits"].double(), ref["logits"].double()
    num = den = dot = na = 0.0
    missing = [k for k in ref["grads"] if k not in got["grads"]]
    per = []
    for k, t in ref["grads"].items():
        if k not in got["grads"]:
            continue
        g, t = got["grads"][k].double(), t.double()
        if g.shape != t.shape:  # (a parameter padded inside the HIP module: _FCNHead's 182 -> 184 channels)
            g = g[tuple(slice(0, n) for n in t.shape)]
        e2, n2, m2, d = float((g - t).norm()) ** 2, float(t.norm()) ** 2, float(g.norm()) ** 2, \
            float((g * t).sum())
        num, den, na, dot = num + e2, den + n2, na + m2, dot + d
        per.append((e2, n2, d / max((m2 * n2) ** 0.5, 1e-300), k))
    per.sort(reverse=True)
    top = [{"tensor": k, "share_of_sq_error": e2 / max(num, 1e-300),
            "share_of_sq_norm": n2 / max(den, 1e-300), "cosine": c} for e2, n2, c, k in per[:6]]
    # arg-max disagreements, adjudicated: a pixel where the two arg-maxes differ is a TIE if the
    # oracle's own top-2 margin there is within the logits bar (1e-3 of the largest |logit|) — the
    # fp32 path reproduces the logits to ~2e-5, so it can only flip classes the oracle itself
    # separates by less than that; anything else is an unexplained mismatch (north_star: "argmax
    # masks bit-identical").
    am, bm = a.argmax(1), b.argmax(1)
    bad = am != bm
    top2 = b.topk(2, dim=1).values
    gap = (top2[:, 0] - top2[:, 1])[bad]
    tie_tol = 1e-3 * float(b.abs().max())
    return dict(
        argmax_mismatch=int(bad.sum()), argmax_pixels=int(bad.numel()),
        argmax_unexplained=int((gap > tie_tol).sum()),
        argmax_largest_margin=float(gap.max()) if gap.numel() else 0.0,
        grad_error_top=top,
        loss_rel=abs(got["loss"] - ref["loss"]) / abs(ref["loss"]),
        logits_maxrel=float((a - b).abs().max() / b.abs().max()),
        logits_l2rel=float((a - b).norm() / b.norm()),
        argmax_agree=float((a.argmax(1) == b.argmax(1)).double().mean()),
        grad_global_rel=(num / den) ** 0.5,
        grad_cosine=dot / (na * den) ** 0.5,
        grad_norm_ratio=(na / den) ** 0.5,
        grad_tensors_missing=len(missing),
        finite=bool(torch.isfinite(a).all()) and na == na)
