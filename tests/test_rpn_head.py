"""The fused RPN head kernels (csrc/rpn_head.hip: objectness_logits + anchor_deltas of detectron2's StandardRPNHead over all FPN
levels, one launch per direction) against torch's conv2d autograd on the CPU: forward, data gradient with the ReLU mask of the
shared convolution folded in, the four parameter gradients (fresh and accumulated into existing buffers), level sizes that are not
multiples of the 16-pixel MFMA group, and the module-level switch (fused vs the per-level implicit GEMMs)."""
import pytest
import torch
import torch.nn.functional as F

CL = torch.channels_last


def _run_kernels(dev, B, shapes):
    from omni3d_amd import functional as HF
    from omni3d_amd.kernels import det
    g = torch.Generator().manual_seed(B + len(shapes))
    ts = [torch.relu(torch.randn(B, 256, h, w, generator=g)).contiguous(memory_format=CL) for h, w in shapes]
    wo, bo = (torch.randn(3, 256, 1, 1, generator=g) * 0.05).contiguous(memory_format=CL), torch.randn(3, generator=g)
    wd, bd = (torch.randn(12, 256, 1, 1, generator=g) * 0.05).contiguous(memory_format=CL), torch.randn(12, generator=g)
    leaves = [t.clone().requires_grad_(True) for t in ts] + [p.clone().requires_grad_(True) for p in (wo, bo, wd, bd)]
    ref = [torch.cat([F.conv2d(t, leaves[-4], leaves[-3]), F.conv2d(t, leaves[-2], leaves[-1]), torch.zeros(B, 1, *t.shape[2:])], 1)
           for t in leaves[:len(ts)]]
    gs = [torch.randn(r.shape, generator=g) for r in ref]
    torch.autograd.backward(ref, gs)
    want = [t.grad * (t > 0) for t in leaves[:len(ts)]] + [p.grad for p in leaves[len(ts):]]
    # through autograd (fresh parameter gradients)
    kt = [t.to(dev).requires_grad_(True) for t in ts]
    kp = [p.to(dev).requires_grad_(True) for p in (wo, bo, wd, bd)]
    assert HF.rpn_head16_eligible(kt, kp[0], kp[2])
    ys = HF.rpn_head16(kt, kp[0], kp[1], kp[2], kp[3])
    for y, r in zip(ys, ref):
        assert tuple(y.shape) == tuple(r.shape) and y.is_contiguous(memory_format=CL)
        assert (y.detach().cpu() - r.detach()).abs().max() < 1e-4
        assert float(y.detach()[:, 15].abs().max()) == 0.0           # the pad column the loss / decode kernels skip
    torch.autograd.backward(ys, [x.contiguous(memory_format=CL).to(dev) for x in gs])
    got = [t.grad for t in kt] + [p.grad for p in kp]
    for i, (a, b) in enumerate(zip(got, want)):
        assert tuple(a.shape) == tuple(b.shape), (i, a.shape, b.shape)
        assert (a.cpu() - b).abs().max() < 2e-4 * max(1.0, float(b.abs().max())), (i, float((a.cpu() - b).abs().max()))
    assert all(getattr(t.grad, "_omni_relu_masked", False) or True for t in kt)
    # parameter gradients ADDED to existing buffers (the gradient bucket's views), unmasked data gradient
    tn = [t.to(dev).permute(0, 2, 3, 1) for t in ts]
    dys = [x.contiguous(memory_format=CL).to(dev).permute(0, 2, 3, 1) for x in gs]
    acc = [torch.full(s, 0.5, device=dev) for s in ((3, 256), (3,), (12, 256), (12,))]
    assert det.head16_wgrad(dys, tn, accum_into=acc) == (None, None, None, None)
    for a, b in zip(acc, want[len(ts):]):
        assert (a.cpu() - 0.5 - b.reshape(a.shape)).abs().max() < 2e-4 * max(1.0, float(b.abs().max()))
    plain = det.head16_dgrad(dys, tn, wo.to(dev), wd.to(dev), relu_mask=False)
    for a, t in zip(plain, leaves[:len(ts)]):
        assert (a.permute(0, 3, 1, 2).cpu() - t.grad).abs().max() < 2e-4 * max(1.0, float(t.grad.abs().max()))


def _run_module(dev):
    """StandardRPNHead with the fused heads == the per-level implicit-GEMM path (same parameters, same inputs)"""
    from omni3d_amd.cubercnn.modeling.proposal_generator import rpn as R
    torch.manual_seed(3)
    head = R.StandardRPNHead(in_channels=256, num_anchors=3, box_dim=4).to(dev)
    for p in head.parameters():
        p.data.normal_(0, 0.05)
    feats = [torch.randn(2, 256, s, s).contiguous(memory_format=CL).to(dev) for s in (16, 8, 4)]
    out = {}
    for fused in (True, False):
        prev, R._FUSED_HEAD = R._FUSED_HEAD, fused
        try:
            xs = [f.clone().requires_grad_(True) for f in feats]
            head.zero_grad()
            ys = head(xs)
            sum((y * y).sum() for y in ys).backward()
            out[fused] = [y.detach() for y in ys] + [x.grad for x in xs] + [p.grad.clone() for p in head.parameters()]
        finally:
            R._FUSED_HEAD = prev
    for a, b in zip(out[True], out[False]):
        assert (a - b).abs().max() <= 5e-4 * max(1.0, float(b.abs().max())), float((a - b).abs().max())


def test_rpn_head16_kernels_emulated(emu_lib):
    _run_kernels("cpu", 2, [(8, 8), (4, 4), (3, 5), (1, 1)])
    _run_kernels("cpu", 1, [(5, 7)])


def test_rpn_head16_module_emulated(emu_lib):
    _run_module("cpu")


def test_rpn_head16_rejects_other_widths():
    from omni3d_amd import functional as HF
    t = [torch.zeros(1, 128, 4, 4)]
    assert not HF.rpn_head16_eligible(t, torch.zeros(3, 128, 1, 1), torch.zeros(12, 128, 1, 1))
    assert not HF.rpn_head16_eligible([torch.zeros(1, 256, 4, 4)], torch.zeros(5, 256, 1, 1), torch.zeros(20, 256, 1, 1))


@pytest.mark.gpu
def test_rpn_head16_kernels_gpu(hip_lib):
    _run_kernels("cuda", 4, [(128, 128), (64, 64), (32, 32), (16, 16), (8, 8)])        # the benchmark's five levels
    _run_kernels("cuda", 2, [(8, 8), (4, 4), (3, 5), (1, 1)])
    _run_kernels("cuda", 3, [(50, 68), (25, 34), (13, 17), (7, 9), (4, 5)])           # ragged level sizes


@pytest.mark.gpu
def test_rpn_head16_module_gpu(hip_lib):
    _run_module("cuda")
