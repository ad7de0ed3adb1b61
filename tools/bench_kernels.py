#!/usr/bin/env python
"""Per-kernel micro-benchmarks on the GPU (HIP-event timed): conv / linear GEMM shapes of
cubercnn_DLA34_FPN at batch 4, 512x512.  Prints TFLOP/s vs the 157.3 TF fp32-MFMA peak."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from omni3d_amd.kernels import conv

B = 4
CONVS = [  # name, H, C, K, R, stride
    ("base 7x7 4->16 @512", 512, 4, 16, 7, 1),
    ("level0 3x3 16->16 @512", 512, 16, 16, 3, 1),
    ("level1 3x3s2 16->32 @512", 512, 16, 32, 3, 2),
    ("l2 3x3s2 32->64 @256", 256, 32, 64, 3, 2),
    ("l2 3x3 64->64 @128", 128, 64, 64, 3, 1),
    ("l2 root 1x1 128->64 @128", 128, 128, 64, 1, 1),
    ("l3 3x3 128->128 @64", 64, 128, 128, 3, 1),
    ("l3 root 1x1 448->128 @64", 64, 448, 128, 1, 1),
    ("l4 3x3 256->256 @32", 32, 256, 256, 3, 1),
    ("l5 3x3 512->512 @16", 16, 512, 512, 3, 1),
    ("fpn/rpn 3x3 256->256 @128", 128, 256, 256, 3, 1),
    ("fpn/rpn 3x3 256->256 @64", 64, 256, 256, 3, 1),
    ("fpn lat 1x1 64->256 @128", 128, 64, 256, 1, 1),
    ("rpn heads 1x1 256->16 @128", 128, 256, 16, 1, 1),
]
LINEARS = [("fc1 2048x12544->1024", 2048, 12544, 1024), ("fc2 2048x1024->1024", 2048, 1024, 1024),
           ("pred 2048x1024->256", 2048, 1024, 256), ("cube fc1 512x12544->1024", 512, 12544, 1024)]


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    # A/B runs: OMNI_TILE / OMNI_SPLITS select the algorithm explicitly through the *_algo entry points (0 = automatic)
    tile, splits = int(os.environ.get("OMNI_TILE", "0")), int(os.environ.get("OMNI_SPLITS", "0"))
    print("tile", tile, "splits", splits)
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    print(f"{'layer':34s} {'GFLOP':>7s} | {'fwd ms':>7s} {'TF':>6s} | {'dgrad':>7s} {'TF':>6s} | {'wgrad':>7s} {'TF':>6s}")
    for name, H, C, K, R, st in CONVS:
        pad = R // 2
        x = torch.randn(B, C, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
        w = (torch.randn(K, C, R, R, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
        y = conv.conv2d_fwd(x, w, None, st, pad)
        dy = torch.randn_like(y)
        gf = 2.0 * B * y.shape[2] * y.shape[3] * K * C * R * R / 1e9
        t1 = timeit(lambda: conv.conv2d_fwd(x, w, None, st, pad, tile=tile, splits=splits))
        t2 = timeit(lambda: conv.conv2d_dgrad(dy, w, (H, H), st, pad, tile=tile, splits=splits))
        t3 = timeit(lambda: conv.conv2d_wgrad(x, dy, (R, R), st, pad, tile=tile))
        for k, t in zip(tot, (t1, t2, t3)):
            tot[k] += t
        print(f"{name:34s} {gf:7.2f} | {t1:7.3f} {gf / t1:6.1f} | {t2:7.3f} {gf / t2:6.1f} | {t3:7.3f} {gf / t3:6.1f}")
    print("direct stem convolution (csrc/stem_conv.hip) vs the implicit GEMM, forward ms")
    for name, C, R in (("base 7x7 4->16 @512", 4, 7), ("level0 3x3 16->16 @512", 16, 3)):
        x = torch.randn(B, C, 512, 512, device="cuda").contiguous(memory_format=torch.channels_last)
        w = (torch.randn(16, C, R, R, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
        gf = 2.0 * B * 512 * 512 * 16 * C * R * R / 1e9
        t1 = timeit(lambda: conv.stem_conv_fwd(x, w))
        t0 = timeit(lambda: conv.conv2d_fwd(x, w, None, 1, R // 2))
        dy = torch.randn(B, 16, 512, 512, device="cuda").contiguous(memory_format=torch.channels_last)
        t2 = timeit(lambda: conv.stem_conv_wgrad(x, dy, R))
        t3 = timeit(lambda: conv.conv2d_wgrad(x, dy, (R, R), 1, R // 2))
        print(f"{name:34s} {gf:7.2f} | fwd stem {t1:7.3f} vs implicit GEMM {t0:7.3f} | wgrad stem {t2:7.3f} vs implicit GEMM {t3:7.3f}")
    from omni3d_amd.kernels import wino
    print("Winograd F(2x2,3x3) path (ms; TF on the DIRECT algorithmic flops)")
    for name, H, CH in (("wino 3x3 256->256 @128", 128, 256), ("wino 3x3 256->256 @64", 64, 256), ("wino 3x3 128->128 @64", 64, 128),
                        ("wino 3x3 256->256 @32", 32, 256), ("wino 3x3 64->64 @128", 128, 64), ("wino 3x3 512->512 @16", 16, 512)):
        x = torch.randn(B, CH, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
        w = (torch.randn(CH, CH, 3, 3, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
        dy = torch.randn_like(x)
        gf = 2.0 * B * H * H * CH * CH * 9 / 1e9
        for tile in (2, 4):
            if H % tile or B * (H // tile) ** 2 < 64:
                continue
            _, V = wino.conv3x3_fwd(x, w, tile=tile)
            t1 = timeit(lambda: wino.conv3x3_fwd(x, w, tile=tile))
            t2 = timeit(lambda: wino.conv3x3_dgrad(dy, w, tile=tile))
            t3 = timeit(lambda: wino.conv3x3_wgrad(V, dy))
            print(f"{name + ' F(%dx%d)' % (tile, tile):34s} {gf:7.2f} | {t1:7.3f} {gf / t1:6.1f} | {t2:7.3f} {gf / t2:6.1f} | {t3:7.3f} {gf / t3:6.1f}")
            Vd, U = wino.transform_input(x, tile), wino.transform_weights(w, tile=tile)[0]
            Mt = wino.gemm_batched(Vd, U)
            dM = wino.transform_dy(dy, tile)
            parts = {"in": lambda: wino.transform_input(x, tile), "w": lambda: wino.transform_weights(w, True, True, tile),
                     "gemm": lambda: wino.gemm_batched(Vd, U), "out": lambda: wino.transform_output(Mt, (B, H, H)),
                     "dy": lambda: wino.transform_dy(dy, tile), "gemm_wgrad": lambda: wino.gemm_batched_wgrad(Vd, dM)}
            print("   parts:", {k: round(timeit(f), 3) for k, f in parts.items()})
    for name, M, C, K in LINEARS:
        x = torch.randn(M, C, device="cuda")
        w = torch.randn(K, C, device="cuda") * 0.02
        dy = torch.randn(M, K, device="cuda")
        gf = 2.0 * M * C * K / 1e9
        t1 = timeit(lambda: conv.linear_fwd(x, w, None))
        t2 = timeit(lambda: conv.linear_dgrad(dy, w))
        t3 = timeit(lambda: conv.linear_wgrad(x, dy))
        print(f"{name:34s} {gf:7.2f} | {t1:7.3f} {gf / t1:6.1f} | {t2:7.3f} {gf / t2:6.1f} | {t3:7.3f} {gf / t3:6.1f}")


if __name__ == "__main__":
    main()
