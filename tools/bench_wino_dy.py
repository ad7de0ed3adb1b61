#!/usr/bin/env python
"""Backward transforms of dy: the one-pass kernel that writes dM and V_dy (wino*_dy_in) against two separate passes
(wino*_dy for dM + wino*_in for V_dy), per map size of the training step; hipGraph-replayed like the step."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omni3d_amd.kernels import wino


def timeit(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 100 * 1e3


for (N, K, H, tile) in ((4, 256, 128, 4), (4, 64, 128, 4), (4, 128, 64, 4), (4, 256, 64, 4), (4, 256, 32, 4), (4, 512, 16, 2), (4, 256, 16, 2)):
    dy = torch.randn(N, K, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    both = timeit(lambda: wino.transform_dy_both(dy, tile))
    t_dm = timeit(lambda: wino.transform_dy(dy, tile))
    t_vd = timeit(lambda: wino.transform_input(dy, tile))
    mb = dy.numel() * 4 / 1e6
    pts = (tile + 2) ** 2 / tile ** 2
    print(f"dy {N}x{K}x{H}x{H} F({tile}): one pass {both:7.1f} us ({(mb + 2 * mb * pts) / both * 1e-3 * 1e3:6.0f} GB/s) | dM {t_dm:6.1f} + V_dy {t_vd:6.1f} = {t_dm + t_vd:7.1f} us", flush=True)
