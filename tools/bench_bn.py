"""BatchNorm forward / backward microbenchmark over the DLA-34 layer shapes (batch 4 @ 512x512), replayed from a hipGraph like
the training step: statistics -> finalize -> apply, three dependent launches (~2.8 us of dispatch each).
    python tools/bench_bn.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omni3d_amd.kernels import bnpool  # noqa: E402

SHAPES = [(4, 512, 16, 16), (4, 256, 32, 32), (4, 128, 64, 64), (4, 64, 128, 128), (4, 32, 256, 256), (4, 16, 512, 512)]


def timeit(fn, iters=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    for shp in SHAPES:
        N, C, H, W = shp
        x = torch.randn(N, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
        dy = torch.randn_like(x)
        gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
        rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
        y, mr, _ = bnpool.bn_fwd(x, gamma, beta, rm, rv, relu=True)
        g = torch.cuda.CUDAGraph()          # replayed as a graph, like the training step
        with torch.cuda.graph(g):
            for _ in range(10):
                y, mr, _ = bnpool.bn_fwd(x, gamma, beta, rm, rv, relu=True)
        tf = timeit(g.replay, 50) / 10
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            for _ in range(10):
                dx = bnpool.bn_bwd(x, dy, y, gamma, mr, relu=True)[0]
        tb = timeit(g2.replay, 50) / 10
        print(f"  {str(shp):22s} {N*C*H*W*4/2**20:7.1f} MB   fwd {tf:7.2f} us   bwd {tb:7.2f} us")


if __name__ == "__main__":
    main()
