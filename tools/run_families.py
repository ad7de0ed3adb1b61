#!/usr/bin/env python
"""Runs one representative layer of every MFMA kernel family of the cubercnn_DLA34_FPN training step (batch 4, 512x512)
through the PRODUCTION dispatch (omni3d_amd.functional: Winograd where the model uses it, direct implicit GEMM elsewhere),
forward + data gradient + weight gradient, so that `rocprofv3 --pmc` passes (tools/pmc_run.py) can attribute MFMA / HBM
counters to each (kernel, grid).  Shapes: SURVEY.md 8(a)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from omni3d_amd import functional as F

B = 4
CONVS = [  # name, H, C, K, R, stride
    ("fpn/rpn 3x3 256->256 @128", 128, 256, 256, 3, 1),
    ("fpn/rpn 3x3 256->256 @64", 64, 256, 256, 3, 1),
    ("l2 3x3 64->64 @128", 128, 64, 64, 3, 1),
    ("l3 3x3 128->128 @64", 64, 128, 128, 3, 1),
    ("l4 3x3 256->256 @32", 32, 256, 256, 3, 1),
    ("l5 3x3 512->512 @16", 16, 512, 512, 3, 1),
    ("l3 3x3s2 64->128 @128", 128, 64, 128, 3, 2),          # round 6: data gradient on dgrad_s2_kernel<64> (all four parity classes per workgroup)
    ("l2 3x3s2 32->64 @256", 256, 32, 64, 3, 2),            # round 6: dgrad_s2_kernel<32>
    ("l3 root 1x1 448->128 @64", 64, 448, 128, 1, 1),
    ("l4 root 1x1 896->256 @32", 32, 896, 256, 1, 1),
    ("fpn lat 1x1 64->256 @128", 128, 64, 256, 1, 1),
    ("level1 3x3s2 16->32 @512", 512, 16, 32, 3, 2),       # round 5: stem_dgrad_s2_kernel / stem_conv_wgrad_kernel<16, 3, 2, 32>
    ("level0 3x3 16->16 @512", 512, 16, 16, 3, 1),
]
LINEARS = [("fc1 2048x12544->1024", 2048, 12544, 1024), ("fc2 2048x1024->1024", 2048, 1024, 1024),
           ("cube fc1 512x12544->1024", 512, 12544, 1024)]
reps = int(os.environ.get("REPS", "3"))
for name, H, C, K, R, st in CONVS:
    x = torch.randn(B, C, H, H, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(K, C, R, R, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    for _ in range(reps):
        with F.wino_weight_scope():
            y = F.conv2d(x, w, None, st, R // 2)
        y.backward(torch.randn_like(y))
for name, M, C, K in LINEARS:
    x = torch.randn(M, C, device="cuda", requires_grad=True)
    w = (torch.randn(K, C, device="cuda") * 0.02).requires_grad_(True)
    # the training step's form: the weight gradient is ADDED into the optimizer's flat bucket by the kernel itself (FlatSGD's direct
    # accumulation) -- for fc1 that is the GEMM engine's TN form with the balanced work split
    w.grad = torch.zeros_like(w)
    w._omni_direct_grad = True
    for _ in range(reps):
        y = F.linear(x, w, None)
        y.backward(torch.randn_like(y))
# the fused RPN head (csrc/rpn_head.hip): both 1x1 heads over the five FPN levels of the benchmark, one launch per direction (HBM-bound)
ts = [torch.relu(torch.randn(B, 256, s, s, device="cuda")).contiguous(memory_format=torch.channels_last).requires_grad_(True) for s in (128, 64, 32, 16, 8)]
wo = (torch.randn(3, 256, 1, 1, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last).requires_grad_(True)
wd = (torch.randn(12, 256, 1, 1, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last).requires_grad_(True)
bo, bd = torch.zeros(3, device="cuda", requires_grad=True), torch.zeros(12, device="cuda", requires_grad=True)
for _ in range(reps):
    ys = F.rpn_head16(ts, wo, bo, wd, bd)
    torch.autograd.backward(ys, [torch.randn_like(y) for y in ys])
# the weight-gradient stream's launch of backward stage 2: 14 Winograd-domain weight-gradient GEMMs in one launch (gemm_tn_multi_kernel)
from omni3d_amd.kernels import wino
STAGE2 = [(36, 4096, 256, 256), (36, 1024, 256, 256), (36, 256, 256, 256), (16, 256, 256, 256)] + [(16, 256, 512, 512)] * 3 + [(36, 256, 256, 256)] * 7
multi = [(torch.randn(b, m, c, device="cuda"), torch.randn(b, m, k, device="cuda")) for b, m, k, c in STAGE2]
for _ in range(reps):
    wino.gemm_batched_wgrad_multi(multi)
torch.cuda.synchronize()
print("done")
