"""The opt-in bf16-split experiment (csrc/gemm_split.hip): time and error against float64 of the Winograd point
GEMMs through the product's fp32-MFMA kernel, the 6-term split and the 3-term split, on the shapes of the step.
usage (GPU box): python tools/bench_gemm_split.py [--json]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from omni3d_amd.kernels import wino  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def run():
    rows = []
    w = torch.randn(4096, 4096, device="cuda")
    for _ in range(200):          # ~0.3 s of load first: the first timed kernel otherwise runs while the clocks still ramp
        w @ w
    torch.cuda.synchronize()
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, B, M, K, C in (("p2 point GEMM (3x3 256->256 @128x128)", 36, 4096, 256, 256), ("p3 point GEMM (@64x64)", 36, 1024, 256, 256),
                             ("DLA level 3 (128->128 @64x64)", 36, 1024, 128, 128), ("DLA level 4 (256->256 @32x32)", 36, 256, 256, 256)):
        V = torch.randn(B, M, C, device="cuda", generator=g)
        U = torch.randn(B, K, C, device="cuda", generator=g) * 0.05
        ref = torch.bmm(V[:4].double(), U[:4].double().transpose(1, 2))          # four of the points in float64
        scale = float(ref.abs().max())
        fl = 2.0 * B * M * K * C
        row = {"shape": f"{name}: {B}x[{M}x{C}]x[{K}x{C}]^T", "gflop": fl / 1e9}
        for key, fn in (("fp32_mfma", lambda: wino.gemm_batched(V, U)), ("split6", lambda: wino.gemm_batched_split(V, U, 6)),
                        ("split3", lambda: wino.gemm_batched_split(V, U, 3))):
            out = fn()
            err = float((out[:4].double() - ref).abs().max()) / scale
            rms = float(((out[:4].double() - ref) ** 2).mean().sqrt()) / scale
            ms = timeit(fn)
            row[key] = {"ms": ms, "tflops_fp32_equivalent": fl / ms / 1e9, "max_err_over_max_ref": err, "rms_err_over_max_ref": rms}
        rows.append(row)
    return rows


if __name__ == "__main__":
    rows = run()
    if "--json" in sys.argv:
        print(json.dumps(rows))
    else:
        for r in rows:
            print(r["shape"])
            for k in ("fp32_mfma", "split6", "split3"):
                v = r[k]
                print(f"   {k:10s} {v['ms'] * 1e3:8.1f} us  {v['tflops_fp32_equivalent']:7.1f} TFLOP/s (fp32-equivalent)   max err {v['max_err_over_max_ref']:.2e}  rms {v['rms_err_over_max_ref']:.2e}")
