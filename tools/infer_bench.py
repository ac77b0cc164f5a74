#!/usr/bin/env python3
"""Inference throughput of the other BASELINE configs on one MI355X (evidence, not the bench
line): C2 DeepLabv3+ mobilenet_v2 @1024x2048 B=1, C5 HRNet-W18-small-v1 @1024x2048 B=16,
C4 PSPNet-resnet101 / C1 FCN-resnet101 eval at 1025x2049 / 480x480.  bf16, random init."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import segmentron_amd  # noqa: E402
from segmentron_amd.config import cfg, reset_cfg  # noqa: E402

CASES = {
    "c2": (["MODEL.MODEL_NAME", "DeepLabV3_Plus", "MODEL.BACKBONE", "mobilenet_v2",
            "MODEL.DEEPLABV3_PLUS.USE_ASPP", "False", "MODEL.DEEPLABV3_PLUS.ENABLE_DECODER", "False"],
           None, 1, 1024, 2048),
    "c5": ([], "configs/cityscapes_hrnet_w18_small_v1.yaml", 16, 1024, 2048),
    "c1": (["MODEL.MODEL_NAME", "FCN", "MODEL.BACKBONE", "resnet101"], None, 1, 480, 480),
    "c4": (["MODEL.MODEL_NAME", "PSPNet", "MODEL.BACKBONE", "resnet101", "MODEL.OUTPUT_STRIDE", "8"],
           None, 1, 1025, 2049),
}


def train_c4():
    """BASELINE C4: PSPNet-resnet101 (OS8, aux loss) train step fwd+bwd, B=2 @1025x2049."""
    reset_cfg()
    cfg.update_from_list(["DATASET.NAME", "cityscape", "TRAIN.BACKBONE_PRETRAINED", "False",
                          "MODEL.MODEL_NAME", "PSPNet", "MODEL.BACKBONE", "resnet101",
                          "MODEL.OUTPUT_STRIDE", "8", "SOLVER.AUX", "True"])
    cfg.PHASE = "train"
    cfg.check_and_freeze()
    segmentron_amd.set_compute_dtype("bf16")
    torch.manual_seed(0)
    model = segmentron_amd.get_segmentation_model().cuda().train()
    B, H, W = 2, 1025, 2049
    x = torch.randn(B, 3, H, W, device="cuda")
    y = torch.randint(0, 19, (B, H, W), device="cuda")

    def step():
        out = model(x)
        loss = torch.nn.functional.cross_entropy(out[0], y) + \
            0.4 * torch.nn.functional.cross_entropy(out[1], y)
        model.zero_grad(set_to_none=True)
        loss.backward()
        return loss
    for _ in range(2):
        loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(json.dumps({"config": "c4-train", "model": "PSPNet", "backbone": "resnet101", "batch": B,
                      "size": [H, W], "ms_per_step": dt * 1e3, "images_per_sec": B / dt,
                      "loss": float(loss), "peak_mem_GB": torch.cuda.max_memory_allocated() / 2 ** 30}),
          flush=True)


def main():
    which = sys.argv[1:] or list(CASES)
    if which == ["c4-train"]:
        return train_c4()
    for tag in which:
        over, yaml_, B, H, W = CASES[tag]
        reset_cfg()
        if yaml_:
            cfg.update_from_file(os.path.join(ROOT, yaml_))
        cfg.update_from_list(["DATASET.NAME", "cityscape", "TRAIN.BACKBONE_PRETRAINED", "False"] + over)
        cfg.PHASE = "test"
        cfg.check_and_freeze()
        segmentron_amd.set_compute_dtype("bf16")
        torch.manual_seed(0)
        model = segmentron_amd.get_segmentation_model().cuda().eval()
        x = torch.randn(B, 3, H, W, device="cuda")
        with torch.no_grad():
            for _ in range(2):
                out = model(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 5
            for _ in range(n):
                out = model(x)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
        # the same forward as ONE HIP graph (segmentron_amd/graph.py)
        from segmentron_amd.graph import GraphedInference
        ginf = GraphedInference(model, x)
        for _ in range(3):
            gout = ginf()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            gout = ginf()
        torch.cuda.synchronize()
        dtg = (time.perf_counter() - t0) / n
        same = bool(torch.equal(gout[0], out[0]))
        print(json.dumps({"config": tag, "model": cfg.MODEL.MODEL_NAME, "backbone": cfg.MODEL.BACKBONE,
                          "batch": B, "size": [H, W], "ms_per_batch": dt * 1e3,
                          "images_per_sec": B / dt, "graph_ms_per_batch": dtg * 1e3,
                          "graph_images_per_sec": B / dtg, "graph_equals_eager": same,
                          "finite": bool(torch.isfinite(out[0]).all()),
                          "out_shape": list(out[0].shape)}), flush=True)
        del ginf, gout
        del model, out, x
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
