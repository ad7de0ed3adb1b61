"""fc1-class GEMMs (box head 2048 rows, cube head 512 rows) on the paths the training step uses; run with OMNI_ENGINE_BAND=1|4|8"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from omni3d_amd.kernels import conv
from tools.bench_engine import timeit
print("OMNI_ENGINE_BAND =", os.environ.get("OMNI_ENGINE_BAND", "(default)"))
for name, M, C, K in (("fc1 box 2048x12544->1024", 2048, 12544, 1024), ("fc1 cube 512x12544->1024", 512, 12544, 1024)):
    x, w, dy = torch.randn(M, C, device="cuda"), torch.randn(K, C, device="cuda") * 0.02, torch.randn(M, K, device="cuda")
    b = torch.randn(K, device="cuda")
    gw = torch.zeros(K, C, device="cuda")
    gf = 2.0 * M * C * K / 1e9
    t1 = timeit(lambda: conv.linear_fwd(x, w, b, True))
    t2 = timeit(lambda: conv.linear_dgrad(dy, w))
    t3 = timeit(lambda: conv.linear_wgrad(x, dy, accum_into=gw))
    print(f"{name:28s} {gf:6.1f} GF | fwd {t1*1e3:6.1f} us {gf/t1/1e3:5.1f} TF | dgrad (incl. transpose) {t2*1e3:6.1f} us {gf/t2/1e3:5.1f} TF | wgrad {t3*1e3:6.1f} us {gf/t3/1e3:5.1f} TF")
