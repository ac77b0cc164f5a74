import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmentron_amd import hip_ops as K
from tools.dw_bench import timeit
dt = torch.bfloat16
for (N, H, W, C) in [(2, 65, 129, 728), (2, 257, 513, 128)]:
    x = torch.randn((N, H, W, C), device="cuda").to(dt)
    w9c = torch.randn((9, C), device="cuda") * 0.3
    s = torch.rand(C, device="cuda") + 0.5
    t = torch.randn(C, device="cuda") * 0.1
    mb = N * H * W * C * 2 / 1e6
    def show(name, us, passes=2):
        print("%-28s %7.1f us  %6.0f GB/s" % (name, us, passes * mb / us * 1e3))
    print((N, H, W, C), "%.1f MB" % mb)
    show("dw pro3 stats", timeit(lambda: K.dwconv(x, w9c, 1, 1, (3, s, t), want_stats=True)))
    show("dw pro3 nostats", timeit(lambda: K.dwconv(x, w9c, 1, 1, (3, s, t), want_stats=False)))
    show("dw pro1 stats", timeit(lambda: K.dwconv(x, w9c, 1, 1, (1, None, None), want_stats=True)))
    show("dw pro0 nostats", timeit(lambda: K.dwconv(x, w9c, 1, 1, None, want_stats=False)))
    show("dw dil2 pro3 stats (generic)", timeit(lambda: K.dwconv(x, w9c, 1, 2, (3, s, t), want_stats=True)))
    show("bn_apply (copy w/ affine)", timeit(lambda: K.bn_apply(x, (3, s, t))))
    y = torch.empty_like(x)
    show("torch copy_", timeit(lambda: y.copy_(x)))
