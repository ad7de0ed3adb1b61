#!/usr/bin/env python
"""fc1-class weight gradient dW (K x C) += dY^T X: plain workgroup order against the XCD-contiguous one (tile code + 32 / + 16 of
omni_conv2d_wgrad_algo), HIP-event timed, accumulate form as in the training step."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omni3d_amd.kernels import conv


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


for M, C, K in ((2048, 12544, 1024), (512, 12544, 1024), (2048, 1024, 1024)):
    x = torch.randn(M, C, 1, 1, device="cuda").contiguous(memory_format=torch.channels_last)
    dy = torch.randn(M, K, 1, 1, device="cuda").contiguous(memory_format=torch.channels_last)
    acc = torch.zeros(K, C, 1, 1, device="cuda").contiguous(memory_format=torch.channels_last)
    gf = 2.0 * M * C * K / 1e9
    res = []
    for name, code in (("plain", 32), ("xcd", 16)):
        t = timeit(lambda: conv.conv2d_wgrad(x, dy, (1, 1), 1, 0, accum_into=acc, tile=code))
        res.append(f"{name}: {t * 1e3:7.1f} us {gf / t:6.1f} TF")
    print(f"{M}x{C}->{K}  {gf:6.2f} GF | " + " | ".join(res), flush=True)
