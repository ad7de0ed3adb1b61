import os, sys
sys.path.insert(0, os.getcwd())
import torch
from omni3d_amd.kernels import conv
def timeit(fn, iters=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
for M, C, K in ((2048, 12544, 1024), (512, 12544, 1024)):
    x = torch.randn(M, C, 1, 1, device="cuda").contiguous(memory_format=torch.channels_last)
    dy = torch.randn(M, K, 1, 1, device="cuda").contiguous(memory_format=torch.channels_last)
    acc = torch.zeros(K, C, 1, 1, device="cuda").contiguous(memory_format=torch.channels_last)
    gf = 2.0 * M * C * K / 1e9
    res = []
    for code in (0, 3+16, 3+32, 1+16, 1+32, 2+16, 2+32):
        for rep in range(2):
            t = timeit(lambda: conv.conv2d_wgrad(x, dy, (1, 1), 1, 0, accum_into=acc, tile=code))
        res.append(f"t{code}: {t*1e3:6.1f}us {gf/t:5.1f}TF")
    print(M, C, K, " | ".join(res), flush=True)
