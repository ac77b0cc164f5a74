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
    # (r06: + its four-wave generation conv_gemm_glds4_kernel<EP, STATS, WM, IM, JN>, 1x1 only)
    glds = [r for r in rows if ("conv_gemm_glds_kernel" in r["Name"] and not _is_kxk(r["Name"]))
            or "conv_gemm_glds4_kernel" in r["Name"]]
    tot = sum(float(r["TotalDurationNs"]) for r in glds)
    calls = sum(int(r["Calls"]) for r in glds)
    out = {"source": os.path.relpath(path, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
           "steps": steps, "glds_launches_per_step": calls / steps,
           "glds_ms_per_step": tot / steps * 1e-6,
           "all_kernels_ms_per_step": sum(float(r["TotalDurationNs"]) for r in rows
                                          if "spin_kernel" not in r["Name"]) / steps * 1e-6}
    # r06: GPU busy fraction with numerator AND denominator from the same graph-replay trace
    # (the old figure divided an eager step's kernel sum by a replayed
    # step's wall time and printed 1.015): sum of kernel durations / (last end - first start) over
    # the middle 60 % of the kernel timeline, from the rocprofv3 kernel trace next to the stats CSV
    trace = [f for f in (path.replace("kernel_stats", "kernel_trace"),
                         os.path.join(os.path.dirname(path), "kernel_trace.csv")) if os.path.exists(f)]
    if len(sys.argv) > 2:
        trace = [sys.argv[2]]
    if trace:
        ev = []
        for r in csv.DictReader(open(trace[0])):
            if "spin_kernel" in r.get("Kernel_Name", ""):
                continue
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        ev.sort()
        if len(ev) > 1000:
            lo, hi = ev[len(ev) // 5][0], ev[len(ev) * 4 // 5][0]
            win = [(a, b) for a, b in ev if lo <= a < hi]
            busy = sum(b - a for a, b in win) / float(max(b for _, b in win) - lo)
            out["graph_busy_frac"] = busy
            out["graph_busy_note"] = ("sum of kernel durations / wall span over the middle 60 %% of the "
                                      "kernel timeline of %s (%d kernels)" % (os.path.basename(trace[0]), len(win)))
    dst = os.path.join(os.path.dirname(os.path.abspath(path)), "rocprof_roofline.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main(sys.argv[1])
