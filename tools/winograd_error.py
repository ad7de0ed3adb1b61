#!/usr/bin/env python
"""fp32 error of the Winograd F(4x4,3x3) path (forward, data gradient, weight gradient) against float64 direct convolution, as
rms error / rms value, next to the same numbers for the direct fp32 kernels.  Runs on the GPU, or (OMNI_EMULATE-style test seam:
`python tools/winograd_error.py cpu`) on the host-compiled kernels."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

dev = sys.argv[1] if len(sys.argv) > 1 else "cuda"
if dev == "cpu":
    from omni3d_amd import lib as L
    L._install_for_tests(L.HipLibrary(os.path.join(ROOT, "tests", "hipemu", "libomni3d_emu.so"), emulated=True))
from omni3d_amd.kernels import conv, wino  # noqa: E402

N, C, K, H = (1, 64, 32, 16) if dev == "cpu" else (4, 256, 256, 64)
g = torch.Generator().manual_seed(0)
x = torch.randn(N, C, H, H, generator=g, dtype=torch.float64)
w = torch.randn(K, C, 3, 3, generator=g, dtype=torch.float64) * 0.05
dy = torch.randn(N, K, H, H, generator=g, dtype=torch.float64)
xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
ref = F.conv2d(xr, wr, None, padding=1)
ref.backward(dy)
cl = lambda t: t.float().contiguous(memory_format=torch.channels_last).to(dev)  # noqa: E731
xk, wk, dyk = cl(x), cl(w), cl(dy)
rms = lambda a, b: float(((a.double().cpu() - b) ** 2).mean().sqrt() / (b ** 2).mean().sqrt())  # noqa: E731
for tile in (4, 2):
    y, V = wino.conv3x3_fwd(xk, wk, None, tile=tile)
    dx = wino.conv3x3_dgrad(dyk, wk, tile=tile)
    dw = wino.conv3x3_wgrad(V, dyk)
    print(f"F({tile}x{tile},3x3): fwd {rms(y, ref.detach()):.2e}  dgrad {rms(dx, xr.grad):.2e}  wgrad {rms(dw, wr.grad):.2e}")
y = conv.conv2d_fwd(xk, wk, None, 1, 1)
dx = conv.conv2d_dgrad(dyk, wk, (H, H), 1, 1)
dw = conv.conv2d_wgrad(xk, dyk, (3, 3), 1, 1)
print(f"direct fp32   : fwd {rms(y, ref.detach()):.2e}  dgrad {rms(dx, xr.grad):.2e}" + (f"  wgrad {rms(dw, wr.grad):.2e}" if dw is not None else ""))
