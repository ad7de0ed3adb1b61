"""Sweep of the batched-GEMM launch choices (omni_gemm_batched_fwd_algo: persistent 128x128 / one 128x128 tile / one 64x64 tile per
workgroup) over the Winograd point-GEMM shapes of the training step; replayed from a hipGraph of 20 launches like the step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from omni3d_amd.kernels import wino

SHAPES = [(36, 4096, 256, 256), (36, 1024, 256, 256), (36, 1024, 128, 128), (36, 256, 256, 256), (36, 4096, 64, 64), (16, 256, 512, 512),
          (16, 1024, 256, 256)]


def timeit(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
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
    return a.elapsed_time(b) / 200 * 1e3


SHAPES += [(36, 64, 256, 256), (16, 256, 256, 256), (36, 5456, 256, 256)]
for (P, M, C, K) in SHAPES:
    V, U = torch.randn(P, M, C, device="cuda"), torch.randn(P, K, C, device="cuda")
    gf = 2.0 * P * M * C * K / 1e9
    res = []
    for algo, wg in ((0, 0), (1, 512), (3, 0), (4, 0), (5, 1024), (5, 768), (5, 512)):
        try:
            t = timeit(lambda: wino.gemm_batched(V, U, algo, wg))
            res.append(f"a{algo}/{wg}: {t:6.1f}us {gf / t * 1e3:5.1f}TF")
        except Exception as e:
            res.append(f"a{algo}/{wg}: err")
    print(f"{str((P, M, C, K)):24s} {gf:6.2f} GF | " + " | ".join(res), flush=True)
    ref = wino.gemm_batched(V, U, 4, 0)
    for wg in (1024, 768, 512, 8):
        if not torch.equal(wino.gemm_batched(V, U, 5, wg), ref):
            print("   !! algo 5 /", wg, "differs from algo 4", flush=True)
    if os.environ.get("SWEEP_WGRAD", "0") != "1":
        continue
    dM = torch.randn(P, M, K, device="cuda")
    res = []
    for algo in (1, 2):
        try:
            t = timeit(lambda: wino.gemm_batched_wgrad(V, dM, algo))
            res.append(f"wgrad a{algo}: {t:6.1f}us {gf / t * 1e3:5.1f}TF")
        except Exception as e:
            res.append(f"wgrad a{algo}: err {type(e).__name__}")
    print(f"{'':24s} {'':9s} | " + " | ".join(res), flush=True)
