"""TEST INFRASTRUCTURE (not product code).  Golden LR sequences from the reference's own schedulers
(segmentron/solver/lr_scheduler.py, imported from /root/reference) for
tests/test_solver.py: tests/golden/lr_schedules.json.  Run in the build container:
    python oracle/gen_golden_lr.py
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("SEGMENTRON_REFERENCE_ROOT", "/root/reference")


def main():
    sys.path.insert(0, REF)
    # torchvision / thop stand-ins for the reference's package-level imports (not in this image)
    sys.path.append(os.path.join(ROOT, "segmentron_amd", "shims"))
    from segmentron.solver import lr_scheduler as ref  # the reference package itself
    cases = [
        dict(kind="poly", max_iters=40, power=0.9, warmup_factor=1.0 / 3, warmup_iters=0, warmup_method="linear"),
        dict(kind="poly", max_iters=50, power=0.9, warmup_factor=1.0 / 3, warmup_iters=7, warmup_method="linear"),
        dict(kind="poly", max_iters=30, power=2.0, warmup_factor=0.1, warmup_iters=5, warmup_method="constant"),
        dict(kind="cosine", max_iters=40, warmup_factor=0.001, warmup_iters=6, warmup_method="linear"),
        dict(kind="step", milestones=[10, 25], gamma=0.1, warmup_factor=0.001, warmup_iters=4, warmup_method="linear"),
    ]
    out = []
    for c in cases:
        p = [torch.nn.Parameter(torch.zeros(1)), torch.nn.Parameter(torch.zeros(1))]
        opt = torch.optim.SGD([{"params": [p[0]], "lr": 0.01}, {"params": [p[1]], "lr": 0.1}], lr=0.01)
        kw = {k: v for k, v in c.items() if k != "kind"}
        sch = {"poly": ref.WarmupPolyLR, "cosine": ref.WarmupCosineLR,
               "step": ref.WarmupMultiStepLR}[c["kind"]](opt, **kw)
        n = c.get("max_iters", 35)
        seq = []
        for _ in range(n):
            seq.append([g["lr"] for g in opt.param_groups])
            opt.step()
            sch.step()
        out.append(dict(case=c, lrs=seq))
    path = os.path.join(ROOT, "tests", "golden", "lr_schedules.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, len(out), "cases")


if __name__ == "__main__":
    main()
