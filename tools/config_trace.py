"""Per-launch timing of one eager step of a bench.py config: every C-ABI call (name + its scalar
arguments = the shape) bracketed by HIP events on the launch stream, written as CSV and summed
by (entry point, shape).  What the per-kernel-name rocprofv3 summary averages away.

    python tools/config_trace.py c5 [out.csv]
"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    key = sys.argv[1] if len(sys.argv) > 1 else "c5"
    out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/trace_%s.csv" % key
    conf = bench.CONFIGS[key]
    import segmentron_amd
    from segmentron_amd import _lib
    from segmentron_amd.config import cfg, reset_cfg
    reset_cfg()
    if conf.get("yaml"):
        cfg.update_from_file(os.path.join(bench.ROOT, conf["yaml"]))
    cfg.update_from_list(conf["over"])
    cfg.PHASE = "train" if conf["train"] else "test"
    cfg.check_and_freeze()
    segmentron_amd.set_compute_dtype("bf16")
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    model = segmentron_amd.get_segmentation_model().to(dev).train(conf["train"])
    images = torch.randn(conf["batch"], 3, conf["h"], conf["w"], device=dev)
    targets = torch.randint(0, 19, (conf["batch"], conf["h"], conf["w"]), device=dev)
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9) if conf["train"] else None

    def step():
        if not conf["train"]:
            with torch.no_grad():
                return model(images)[0]
        outs = model(images)
        loss = sum(torch.nn.functional.cross_entropy(o, targets, ignore_index=-1) for o in outs)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    for _ in range(3):
        step()
    torch.cuda.synchronize()

    klass = _lib._Lib
    orig = klass.call
    rec = []

    def traced(self, name, *args):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(self, name, *args)
        e1.record()
        rec.append((name, tuple(a for a in args if isinstance(a, (int, float)) and abs(a) < (1 << 31)),
                    e0, e1))

    klass.call = traced
    try:
        # keep the queue full: a long spin kernel first, so the events measure device time
        torch.cuda._sleep(200_000_000)
        step()
        torch.cuda.synchronize()
    finally:
        klass.call = orig
    rows = [(n, a, e0.elapsed_time(e1) * 1e3) for n, a, e0, e1 in rec]
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    with open(out, "w", newline="") as f:
        wr = csv.writer(f)
        wr.writerow(["idx", "us", "entry", "scalars"])
        for i, (n, a, us) in enumerate(rows):
            wr.writerow([i, "%.1f" % us, n, " ".join(str(x) for x in a)])
    agg = {}
    for n, a, us in rows:
        t = agg.setdefault((n, a), [0.0, 0])
        t[0] += us
        t[1] += 1
    tot = sum(us for _, _, us in rows)
    print("%s: %d C-ABI calls, %.3f ms between their events" % (key, len(rows), tot * 1e-3))
    for (n, a), (us, k) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
        print("%9.1f us  x%-3d %-28s %s" % (us, k, n, " ".join(str(x) for x in a)))


if __name__ == "__main__":
    main()
