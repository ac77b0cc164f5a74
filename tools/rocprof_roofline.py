#!/usr/bin/env python3
"""profiles/rocprof_roofline.json from a `rocprofv3 --kernel-trace --stats` CSV of `bench.py`.

    python tools/rocprof_roofline.py profiles/r04x_graph_kernel_stats.csv

bench.py divides the dominant kernel's algorithmic FLOPs per step (counted live) by the per-step
time written here -> `roofline.frac_rocprof` next to the HIP-event figure (VERDICT r03 #12b).
Steps = calls of the once-per-step `ce_fwd_kernel` (graph replays and eager steps alike)."""
import csv
import json
import os
import sys


def _is_kxk(name):
    """conv_gemm_glds_kernel<EP, STATS, KXK, IMS>: third template argument"""
    args = name[name.find("<") + 1:name.rfind(">")].replace(" ", "").split(",")
    return len(args) >= 3 and args[2] in ("true", "1")


def main(path):
    rows = list(csv.DictReader(open(path)))
    steps = sum(int(r["Calls"]) for r in rows if "ce_fwd_kernel" in r["Name"])
    if steps == 0:
        sys.exit("no ce_fwd_kernel launches in %s" % path)
    # the direct-to-LDS 1x1 GEMM (its KxK instances excluded: bench.py counts the FLOPs of the
    # 1x1 launches it times)
    glds = [r for r in rows if "conv_gemm_glds_kernel" in r["Name"] and not _is_kxk(r["Name"])]
    tot = sum(float(r["TotalDurationNs"]) for r in glds)
    calls = sum(int(r["Calls"]) for r in glds)
    out = {"source": os.path.relpath(path, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
           "steps": steps, "glds_launches_per_step": calls / steps,
           "glds_ms_per_step": tot / steps * 1e-6,
           "all_kernels_ms_per_step": sum(float(r["TotalDurationNs"]) for r in rows
                                          if "spin_kernel" not in r["Name"]) / steps * 1e-6}
    dst = os.path.join(os.path.dirname(os.path.abspath(path)), "rocprof_roofline.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main(sys.argv[1])
