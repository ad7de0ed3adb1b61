"""fc1 data gradient: W transposed + the engine's NT form (production) against the engine's NN form reading W as it is"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from omni3d_amd.kernels import conv, gemm as G
from tools.bench_engine import timeit
for name, M in (("box 2048", 2048), ("cube 512", 512)):
    C, K = 12544, 1024
    w, dy = torch.randn(K, C, device="cuda") * 0.02, torch.randn(M, K, device="cuda")
    gf = 2.0 * M * C * K / 1e9
    t0 = timeit(lambda: conv.linear_dgrad(dy, w))
    ref = conv.linear_dgrad(dy, w)
    for tile, splits in ((2, G.BALANCED), (2, 1), (1, 1), (1, G.BALANCED)):
        try:
            t = timeit(lambda: G.gemm(dy, w, G.NN, tile=tile, splits=splits))
            out = G.gemm(dy, w, G.NN, tile=tile, splits=splits)
            err = float((out - ref).abs().max())
            print(f"{name}: production {t0*1e3:6.1f} us ({gf/t0/1e3:5.1f} TF) | engine NN tile {tile} splits {splits}: {t*1e3:6.1f} us ({gf/t/1e3:5.1f} TF) max diff {err:.2e}")
        except Exception as e:
            print(name, tile, splits, "failed", str(e)[:80])
