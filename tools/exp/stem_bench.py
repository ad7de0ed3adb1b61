import os, sys
sys.path.insert(0, os.getcwd())
import torch
from omni3d_amd.kernels import conv
from tools.bench_kernels import timeit
B = 4
for name, C, K, R, st in (("base 7x7 4->16 @512", 4, 16, 7, 1), ("level0 3x3 16->16 @512", 16, 16, 3, 1), ("level1 3x3s2 16->32 @512", 16, 32, 3, 2)):
    x = torch.randn(B, C, 512, 512, device="cuda").contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, K, 512 // st, 512 // st, device="cuda").contiguous(memory_format=torch.channels_last)
    w = (torch.randn(K, C, R, R, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    gf = 2.0 * B * (512 // st) ** 2 * K * C * R * R / 1e9
    t2 = timeit(lambda: conv.stem_conv_wgrad(x, dy, R, stride=st), 20)
    t3 = timeit(lambda: conv.conv2d_wgrad(x, dy, (R, R), st, R // 2), 20)
    t1 = timeit(lambda: conv.conv2d_fwd(x, w, None, st, R // 2), 20)
    t4 = timeit(lambda: conv.conv2d_dgrad(dy, w, (512, 512), st, R // 2), 20) if C == 16 else 0.0
    print(f"{name:30s} {gf:6.2f} GF | wgrad stem {t2*1e3:7.1f} us vs implicit GEMM {t3*1e3:7.1f} us | implicit fwd {t1*1e3:7.1f} dgrad {t4*1e3:7.1f}")
