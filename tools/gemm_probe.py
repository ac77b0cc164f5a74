"""How does the 256x128 GEMM's time scale with the number of resident blocks / K?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmentron_amd import hip_ops as K
from tools.gemm_bench import timeit
dt = torch.bfloat16
ONLY = int(os.environ.get("PROBE_ONLY", "-1"))
ITERS = int(os.environ.get("PROBE_ITERS", "30"))
for idx, (M_h, C, O) in enumerate([(65, 728, 728), (33, 728, 728), (16, 728, 728), (65, 1456, 728), (65, 2912, 728), (65, 728, 1456), (65, 728, 384)]):
    if ONLY >= 0 and idx != ONLY:
        continue
    N, H, W = 2, M_h, 129
    x = torch.randn((N, H, W, C), device="cuda").to(dt)
    w = (torch.randn((O, C), device="cuda") * 0.05).to(dt)
    t = timeit(lambda: K.conv_gemm(x, w, O, 1, 1, 1, 0, 1), ITERS)
    M = N * H * W
    blocks = ((M + 255) // 256) * ((O + 127) // 128)
    flop = 2.0 * M * C * O
    print("M=%6d K=%5d N=%5d blocks=%4d slabs=%3d  %7.1f us  %6.0f TF  (%.2f us/slab)"
          % (M, C, O, blocks, (C + 63) // 64, t * 1e6, flop / t / 1e12, t * 1e6 / ((C + 63) // 64)))
