#!/usr/bin/env python
"""fc1-class weight gradient dW (K x C) += dY^T X: plain workgroup order against the XCD-contiguous one (tile code + 32 / + 16 of
omni_conv2d_wgrad_algo) and against the LDS-DMA engine's TN form (plain 128x128 tiles, and the balanced split: whole tiles per
workgroup + one part of the left-over tiles each), HIP-event timed, accumulate form as in the training step.  Second table: the
fc1 data gradient (transpose of W + NT engine) with and without the balanced split."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omni3d_amd.kernels import conv, gemm as G


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
    x2, dy2, acc2 = x.view(M, C), dy.view(M, K), acc.view(K, C)
    for name, kw in (("engine t2", dict(tile=2)), ("engine t2 balanced", dict(tile=2, splits=G.BALANCED)), ("engine t1 balanced", dict(tile=1, splits=G.BALANCED))):
        t = timeit(lambda: G.gemm(dy2, x2, G.TN, out=acc2, accumulate=True, **kw))
        res.append(f"{name}: {t * 1e3:7.1f} us {gf / t:6.1f} TF")
    print(f"wgrad {M}x{C}->{K}  {gf:6.2f} GF | " + " | ".join(res), flush=True)
    w = torch.randn(K, C, device="cuda") * 0.02
    res = []
    for name, kw in (("transpose + NT t2", dict(tile=2)), ("transpose + NT t2 balanced", dict(tile=2, splits=G.BALANCED)), ("transpose + NT t1 balanced", dict(tile=1, splits=G.BALANCED))):
        t = timeit(lambda: G.gemm(dy2, G.transpose2d(w), G.NT, **kw))
        res.append(f"{name}: {t * 1e3:7.1f} us {gf / t:6.1f} TF")
    print(f"dgrad {M}x{K}->{C}  {gf:6.2f} GF | " + " | ".join(res), flush=True)
