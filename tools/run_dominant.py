#!/usr/bin/env python
"""Runs only the dominant kernel of the training step (3x3 256->256 conv on the 128x128 map, batch 4) so that
rocprofv3 --pmc passes can attribute HBM traffic counters to it (see profiles/README.md)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from omni3d_amd.kernels import conv

x = torch.randn(4, 256, 128, 128, device="cuda").contiguous(memory_format=torch.channels_last)
w = (torch.randn(256, 256, 3, 3, device="cuda") * 0.02).contiguous(memory_format=torch.channels_last)
dy = torch.randn(4, 256, 128, 128, device="cuda").contiguous(memory_format=torch.channels_last)
for _ in range(5):
    conv.conv2d_fwd(x, w, None, 1, 1)
    conv.conv2d_dgrad(dy, w, (128, 128), 1, 1)
    conv.conv2d_wgrad(x, dy, (3, 3), 1, 1)
# the Winograd batched GEMM of the same layer (gemm_nt_persistent_kernel)
from omni3d_amd.kernels import wino
V, U = wino.transform_input(x, 4), wino.transform_weights(w, tile=4)[0]     # F(4x4,3x3): 36 x [4096 x 256] * [256 x 256]^T
dM = wino.transform_dy(dy, 4)
for _ in range(5):
    wino.gemm_batched(V, U)
    wino.gemm_batched_wgrad(V, dM)
torch.cuda.synchronize()
print("done")
