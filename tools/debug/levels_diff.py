import os, sys
sys.path.insert(0, os.getcwd())
import torch, torch.nn.functional as F
from omni3d_amd import functional as HF
g = torch.Generator().manual_seed(3)
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
N, C, K = 4, 256, 256
sizes = [(128, 128), (64, 64), (32, 32), (16, 16), (8, 8)]
xs = [torch.randn(N, C, h, w, generator=g) for h, w in sizes]
w = torch.randn(K, C, 3, 3, generator=g) * 0.05
b = torch.randn(K, generator=g) * 0.1
dys = [torch.randn(N, K, h, w_, generator=g) for h, w_ in sizes]
xr = [x.clone().cuda().requires_grad_(True) for x in xs]
wr, br = w.clone().cuda().requires_grad_(True), b.clone().cuda().requires_grad_(True)
ref = [F.relu(F.conv2d(x, wr, br, padding=1)) for x in xr]
torch.autograd.backward(ref, [d.cuda() for d in dys])
xd = [cl(x).cuda().requires_grad_(True) for x in xs]
wd, bd = cl(w).cuda().requires_grad_(True), b.cuda().requires_grad_(True)
with HF.wino_weight_scope():
    ys = HF.conv3x3_levels(xd, wd, bd, relu=True)
torch.autograd.backward(ys, [cl(d).cuda() for d in dys])
for l in range(len(sizes)):
    print(sizes[l], "y", float((ys[l] - ref[l]).abs().max()), "dx", float((xd[l].grad - xr[l].grad).abs().max()), "scale", float(xr[l].grad.abs().max()))
print("dw", float((wd.grad - wr.grad).abs().max()), float(wr.grad.abs().max()), "db", float((bd.grad - br.grad).abs().max()))
