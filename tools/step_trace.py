"""One eager C3 train step under torch.profiler: every GPU kernel in launch order with its
duration and grid, written as CSV (gpurun_out/step_trace.csv).  Used to attribute time to the
entry flow / decoder shapes that the per-kernel-name rocprofv3 summary averages away."""
import os, sys, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity


def build_train_step(dev, batch, hw, dtype):
    import segmentron_amd
    from segmentron_amd.config import cfg, reset_cfg
    reset_cfg()
    cfg.update_from_list(bench.C3)
    cfg.PHASE = "train"
    cfg.check_and_freeze()
    segmentron_amd.set_compute_dtype(dtype)
    torch.manual_seed(0)
    model = segmentron_amd.get_segmentation_model().to(dev).train()
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    images = torch.randn(batch, 3, *hw, device=dev)
    targets = torch.randint(0, 19, (batch, *hw), device=dev)

    def step():
        out = model(images)
        loss = torch.nn.functional.cross_entropy(out[0], targets, ignore_index=-1)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss
    return model, step

def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/step_trace.csv"
    dev = torch.device("cuda:0")
    model, step = build_train_step(dev, 2, (1025, 2049), "bf16")
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        step()
        torch.cuda.synchronize()
    evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    evs.sort(key=lambda e: e.time_range.start)
    with open(out, "w", newline="") as f:
        wr = csv.writer(f)
        wr.writerow(["idx", "start_us", "dur_us", "name"])
        t0 = evs[0].time_range.start if evs else 0
        for i, e in enumerate(evs):
            wr.writerow([i, "%.1f" % (e.time_range.start - t0), "%.1f" % e.device_time, e.name[:110]])
    print("kernels", len(evs), "sum ms", sum(e.device_time for e in evs) / 1e3)

if __name__ == "__main__":
    main()
