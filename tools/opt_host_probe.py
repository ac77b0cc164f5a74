import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
import segmentron_amd
from segmentron_amd.config import cfg, reset_cfg
from segmentron_amd.solver.optimizer import FusedSGD
reset_cfg(); cfg.update_from_list(bench.C3); cfg.PHASE = "train"; cfg.check_and_freeze()
segmentron_amd.set_compute_dtype("bf16")
m = segmentron_amd.get_segmentation_model().cuda().train()
for p in m.parameters(): p.grad = torch.zeros_like(p)
for name, opt in (("FusedSGD", FusedSGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)),
                  ("torch fused", torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, fused=True)),
                  ("torch foreach", torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4))):
    for _ in range(3): opt.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): opt.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-14s host %.2f ms/step, incl. GPU %.2f ms/step" % (name, (t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
