#!/usr/bin/env python3
"""Per-step budget of the bf16 C3 train step from a `rocprofv3 --kernel-trace --stats` CSV of
bench.py (profiles/r05*_graph_kernel_stats.csv): kernels grouped by class, ms and launches per
step.  The bench command also runs a few fp32 steps (extra legs): kernels instantiated for float
are reported separately and not counted in the bf16 step.

    python tools/step_budget.py profiles/r05e_graph_kernel_stats.csv
"""
import csv
import re
import sys

CLASSES = [
    ("forward / data-gradient GEMM, direct-to-LDS (conv_gemm_glds)", r"conv_gemm_glds_kernel|conv_gemm_glds4_kernel|conv_gemm_g4_kernel"),
    ("forward / data-gradient convolutions, other (first-generation GEMM, px256, direct 3x3)",
     r"conv_gemm_fwd_kernel|conv_gemm_px256_kernel|conv3x3_direct_kernel"),
    ("weight-gradient GEMM", r"conv_wgrad_glds_kernel|conv_gemm_wgrad_kernel|conv3x3_wgrad_direct"),
    ("depthwise forward", r"dwconv_slide_fwd|dwconv_tiled_kernel|dwconv_tiled_s2_kernel|dwconv_row_kernel<[^>]*false>|dwconv_kernel"),
    ("depthwise fused backward", r"dwconv_slide_bwd|dwconv_bwd|dwconv_row_kernel<[^>]*true>|dwconv_wgrad|dwconv_dgrad"),
    ("BatchNorm element-wise passes (apply, backward apply / reduce, n-ary gradient sum)",
     r"bn_apply_kernel|bn_bwd_apply_kernel|bn_bwd_reduce_kernel|sum_n_kernel"),
    ("finalize / fold / column-sum micro-kernels",
     r"finalize|fold_|colsum|bn_eval_affine|bn_bwd_small|bn_moments"),
    ("loss, metric, resampling, pooling", r"ce_fwd|ce_bwd|metric|bilinear|upsample|nearest|maxpool|avgpool|nchw_to_nhwc"),
    ("optimizer + weight packing", r"sgd_multi_tensor|pack_multi"),
]


def main(path):
    rows = list(csv.DictReader(open(path)))
    steps_all = sum(int(r["Calls"]) for r in rows if "ce_fwd_kernel" in r["Name"])
    steps_f32 = sum(int(r["Calls"]) for r in rows if "ce_fwd_kernel<float" in r["Name"])
    steps = steps_all - steps_f32
    acc = {name: [0.0, 0.0] for name, _ in CLASSES}
    acc["torch glue (casts, fills, masks, adds)"] = [0.0, 0.0]
    f32 = [0.0, 0.0]
    for r in rows:
        n = r["Name"]
        if "spin_kernel" in n:
            continue
        ms, calls = float(r["TotalDurationNs"]) * 1e-6, int(r["Calls"])
        # (finalize / column-sum kernels are templated on the PARTIAL type, float in every step)
        if re.search(r"<float[,>]", n) and "seg::" in n and not re.search(r"finalize|colsum", n):
            f32[0] += ms
            f32[1] += calls
            continue
        for name, pat in CLASSES:
            if re.search(pat, n):
                acc[name][0] += ms
                acc[name][1] += calls
                break
        else:
            acc["torch glue (casts, fills, masks, adds)"][0] += ms
            acc["torch glue (casts, fills, masks, adds)"][1] += calls
    tot = sum(v[0] for v in acc.values()) / steps
    print("%s: %d bf16 steps (+ %d fp32 steps of the extra legs: %.1f ms of float kernels not counted)"
          % (path, steps, steps_f32, f32[0]))
    print("| class | ms / step | launches / step | share |")
    print("|---|---|---|---|")
    for name, (ms, calls) in acc.items():
        print("| %s | %.2f | %.0f | %.0f %% |" % (name, ms / steps, calls / steps, 100 * ms / steps / tot))
    print("| **sum** | **%.2f** | **%.0f** | |" % (tot, sum(v[1] for v in acc.values()) / steps))


if __name__ == "__main__":
    main(sys.argv[1])
