#!/usr/bin/env python3
"""Per-LAYER comparison of two rocprofv3 kernel traces of the same bench command under two builds
of the C-ABI library (tools/lab/prof_ab.sh): the captured step launches the same sequence of
kernels in both, so the i-th dispatch of a step is the same layer — durations are averaged per
ordinal over the replayed steps and printed for the ordinals whose kernel name matches `pattern`
in either trace, grouped by (kernel A, grid A) -> (kernel B, grid B).

    python tools/lab/trace_ab.py A_kernel_trace.csv B_kernel_trace.csv [pattern]
"""
import csv
import sys
from collections import defaultdict


def steps(path):
    ev = []
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "spin_kernel" in n:
            continue
        grid = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, grid))
    ev.sort()
    # a step = the dispatches between two "first ce_fwd_kernel launches after an optimizer launch"
    # (models with auxiliary heads run several loss kernels per step); keep the most common length
    cuts, seen_opt = [], True
    for i, e in enumerate(ev):
        if "sgd_multi_tensor_kernel" in e[2]:
            seen_opt = True
        elif "ce_fwd_kernel" in e[2] and seen_opt:
            cuts.append(i)
            seen_opt = False
    segs = [ev[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    lens = defaultdict(int)
    for s in segs:
        lens[len(s)] += 1
    L = max(lens, key=lens.get)
    segs = [s for s in segs if len(s) == L]
    sig = defaultdict(int)
    for s in segs:
        sig[tuple(e[2] for e in s)] += 1
    best = max(sig, key=sig.get)
    segs = [s for s in segs if tuple(e[2] for e in s) == best]
    return segs[len(segs) // 4:]  # (the first replays: clocks still settling)


def short(n):
    n = n.replace("void seg::", "").replace("(seg::ConvGemmArgs)", "")
    return n[:52]


def main():
    a, b = steps(sys.argv[1]), steps(sys.argv[2])
    pat = sys.argv[3] if len(sys.argv) > 3 else "conv_gemm_glds"
    L = len(a[0])
    if len(b[0]) != L:
        sys.exit("different launch counts per step: %d vs %d" % (L, len(b[0])))
    print("%d / %d steady steps of %d launches" % (len(a), len(b), L))
    groups = defaultdict(lambda: [0, 0.0, 0.0])
    ta = tb = 0.0
    for i in range(L):
        da = sum(s[i][1] - s[i][0] for s in a) / len(a) * 1e-3
        db = sum(s[i][1] - s[i][0] for s in b) / len(b) * 1e-3
        ta += da
        tb += db
        if pat in a[0][i][2] or pat in b[0][i][2]:
            g = groups[(short(a[0][i][2]), a[0][i][3], short(b[0][i][2]), b[0][i][3])]
            g[0] += 1
            g[1] += da
            g[2] += db
    print("all kernels: %.3f -> %.3f ms per step" % (ta * 1e-3, tb * 1e-3))
    sa = sb = 0.0
    for rank, (k, (n, da, db)) in enumerate(sorted(groups.items(), key=lambda kv: -kv[1][1])):
        sa += da
        sb += db
        if rank >= int(__import__("os").environ.get("TRACE_AB_TOP", "1000")):
            continue
        print("%3d x  %-52s grid %-8s %8.2f us  ->  %-52s grid %-8s %8.2f us  (%+.1f %%)"
              % (n, k[0], k[1], da / n, k[2], k[3], db / n, (db / da - 1) * 100))
    print("matching launches: %.3f -> %.3f ms per step" % (sa * 1e-3, sb * 1e-3))


if __name__ == "__main__":
    main()
