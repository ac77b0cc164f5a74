"""Reference point (NOT on the product path): what torch.mm (hipBLASLt/rocBLAS) reaches on the
dominant GEMM shapes, next to seg_conv_gemm_fwd on the same tensors."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmentron_amd import hip_ops as K
from tools.gemm_bench import timeit
dt = torch.bfloat16
for (M, Kd, N) in [(16770, 728, 728), (16770, 1536, 2048), (263682, 304, 256), (1051650, 128, 128)]:
    x = torch.randn((M, Kd), device="cuda").to(dt)
    w = (torch.randn((N, Kd), device="cuda") * 0.05).to(dt)
    y = torch.empty((M, N), device="cuda", dtype=dt)
    t_lib = timeit(lambda: torch.mm(x, w.t(), out=y), 30)
    x4 = x.view(1, 1, M, Kd)
    t_seg = timeit(lambda: K.conv_gemm(x4, w, N, 1, 1, 1, 0, 1), 30)
    fl = 2.0 * M * Kd * N
    print("M=%7d K=%5d N=%5d  torch.mm %7.1f us %6.0f TF | seg_conv_gemm_fwd %7.1f us %6.0f TF"
          % (M, Kd, N, t_lib * 1e6, fl / t_lib / 1e12, t_seg * 1e6, fl / t_seg / 1e12))
