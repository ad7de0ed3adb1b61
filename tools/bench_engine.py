#!/usr/bin/env python
"""GEMM-shaped launches of the training step: round-1 tile engine (csrc/conv_gemm.hip) vs the round-2 engine
(csrc/gemm_engine.hip), HIP-event timed.  TFLOP/s against the 157.3 fp32-MFMA peak."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from omni3d_amd.kernels import conv, gemm as G, wino


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def row(name, gf, olds, news):
    s = f"{name:44s} {gf:7.2f} GF | old " + " ".join(f"{k} {t:.3f}ms {gf / t:5.1f}TF" for k, t in olds)
    s += " | new " + " ".join(f"{k} {t:.3f}ms {gf / t:5.1f}TF" for k, t in news)
    print(s, flush=True)


def main():
    dev = "cuda"
    # ---- Winograd point GEMMs (NT) and their weight gradients (TN)
    for name, b, M, N, K in (("wino p2 36x[4096x256x256]", 36, 4096, 256, 256), ("wino p3 36x[1024x256x256]", 36, 1024, 256, 256),
                             ("wino F(2,3) p4 16x[1024x256x256]", 16, 1024, 256, 256), ("wino l3 36x[1024x128x128]", 36, 1024, 128, 128),
                             ("wino l2 36x[4096x64x64]", 36, 4096, 64, 64)):
        V, U = torch.randn(b, M, K, device=dev), torch.randn(b, N, K, device=dev)
        gf = 2.0 * b * M * N * K / 1e9
        olds = [("auto", timeit(lambda: wino.gemm_batched(V, U)))]
        news = [(f"t{t}", timeit(lambda: G.gemm(V, U, G.NT, tile=t))) for t in (1, 2)]
        row(name + " NT", gf, olds, news)
        dM = torch.randn(b, M, N, device=dev)
        olds = [("auto", timeit(lambda: wino.gemm_batched_wgrad(V, dM)))]
        out = torch.empty(b, N, K, device=dev)
        news = [(f"t{t}s{s}", timeit(lambda: G.gemm(dM, V, G.TN, out=out, tile=t, splits=s))) for t, s in ((1, 1), (2, 1), (2, 4), (1, 4))]
        row(name + " TN(wgrad)", gf, olds, news)
    # ---- FC layers
    for name, M, C, K in (("fc1 box 2048x12544->1024", 2048, 12544, 1024), ("fc1 cube 512x12544->1024", 512, 12544, 1024),
                          ("fc2 2048x1024->1024", 2048, 1024, 1024), ("pred 2048x1024->256", 2048, 1024, 256)):
        x, w, dy = torch.randn(M, C, device=dev), torch.randn(K, C, device=dev) * 0.02, torch.randn(M, K, device=dev)
        gf = 2.0 * M * C * K / 1e9
        row(name + " fwd NT", gf, [("auto", timeit(lambda: conv.linear_fwd(x, w, None)))],
            [(f"t{t}s{s}", timeit(lambda: G.gemm(x, w, G.NT, tile=t, splits=s))) for t, s in ((1, 1), (1, 4), (2, 2), (2, 8))])
        row(name + " dgrad NN", gf, [("auto", timeit(lambda: conv.linear_dgrad(dy, w)))],
            [(f"t{t}s{s}", timeit(lambda: G.gemm(dy, w, G.NN, tile=t, splits=s))) for t, s in ((1, 1), (2, 1), (2, 2))])
        out = torch.empty(K, C, device=dev)
        row(name + " wgrad TN", gf, [("auto", timeit(lambda: conv.linear_wgrad(x, dy)))],
            [(f"t{t}s{s}", timeit(lambda: G.gemm(dy, x, G.TN, out=out, tile=t, splits=s))) for t, s in ((1, 1), (2, 1), (2, 2), (1, 2))])
    # ---- 1x1 convolutions as GEMMs (NHWC activations are the row-major A operand)
    for name, P, C, K in (("fpn lateral 1x1 64->256 @128 (65536 px)", 65536, 64, 256), ("root 1x1 448->128 @64 (16384 px)", 16384, 448, 128),
                          ("root 1x1 896->256 @32 (4096 px)", 4096, 896, 256), ("root 1x1 128->64 @128 (65536 px)", 65536, 128, 64)):
        x, w, dy = torch.randn(P, C, device=dev), torch.randn(K, C, device=dev) * 0.05, torch.randn(P, K, device=dev)
        gf = 2.0 * P * C * K / 1e9
        row(name + " fwd NT", gf, [("auto", timeit(lambda: conv.linear_fwd(x, w, None)))],
            [(f"t{t}", timeit(lambda: G.gemm(x, w, G.NT, tile=t))) for t in (1, 2)])
        row(name + " dgrad NN", gf, [("auto", timeit(lambda: conv.linear_dgrad(dy, w)))],
            [(f"t{t}", timeit(lambda: G.gemm(dy, w, G.NN, tile=t))) for t in (1, 2)])
        out = torch.empty(K, C, device=dev)
        row(name + " wgrad TN", gf, [("auto", timeit(lambda: conv.linear_wgrad(x, dy)))],
            [(f"t{t}s{s}", timeit(lambda: G.gemm(dy, x, G.TN, out=out, tile=t, splits=s))) for t, s in ((2, 16), (2, 64), (1, 64), (2, 128))])


if __name__ == "__main__":
    main()
