"""Isolated timing of the stride-2 data gradients of DLA-34 at the benchmark's size: fused four-class kernel (csrc/dgrad_s2.hip) vs
the generic kernel's grid.z parity classes.  usage (GPU box): python tools/bench_s2_dgrad.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from omni3d_amd.kernels import conv  # noqa: E402


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


CL = torch.channels_last
for N, C, H, K in ((4, 32, 256, 64), (4, 64, 128, 128), (4, 128, 64, 256), (4, 256, 32, 512)):
    OH = (H - 1) // 2 + 1
    dy = torch.randn(N, K, OH, OH, device="cuda").contiguous(memory_format=CL)
    w = (torch.randn(K, C, 3, 3, device="cuda") * 0.05).contiguous(memory_format=CL)
    fl = 2.0 * N * OH * OH * K * 9 * C
    res = {}
    for name, flag, minw in (("fused", True, 1), ("generic", False, 1)):
        conv._S2_DGRAD, conv._S2_DGRAD_MIN_WGS = flag, minw
        us = timeit(lambda: conv.conv2d_dgrad(dy, w, (H, H), 2, 1))
        res[name] = us
    print(f"3x3/s2 {C:3d}->{K:3d} dx {H}x{H}: fused {res['fused']:7.1f} us ({fl / res['fused'] / 1e6 / 157.3:.2f} of peak)   generic {res['generic']:7.1f} us "
          f"({fl / res['generic'] / 1e6 / 157.3:.2f})", flush=True)
