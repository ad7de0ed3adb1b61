"""Depthwise convolution kernels (csrc/depthwise.hip) against torch, and the MNASNet-1.0 + FPN bottom-up
(configs/cubercnn_mnasnet_FPN.yaml) against the REFERENCE's own wrapper (cubercnn/modeling/backbone/mnasnet.py under
oracle/ref_harness.py) over the oracle's restatement of torchvision's mnasnet1_0 (un-vendored: that restatement is pinned only
by torchvision's published parameter count)."""
import os

import pytest
import torch
import torch.nn.functional as F

REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.exists(REF), reason="needs the reference checkout (build container only)")

DW_CASES = [(2, 13, 17, 8, 3, 1), (2, 13, 17, 8, 3, 2), (1, 16, 12, 72, 5, 2), (3, 9, 9, 240, 5, 1), (1, 4, 4, 1152, 5, 2), (1, 33, 31, 32, 3, 1)]


def _run_dw(dev):
    from omni3d_amd import functional as HF
    g = torch.Generator().manual_seed(2)
    for N, H, W, C, R, stride in DW_CASES:
        x = torch.randn(N, C, H, W, generator=g)
        w = torch.randn(C, 1, R, R, generator=g) * 0.3
        xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        yr = F.conv2d(xr, wr, None, stride, R // 2, 1, C)
        go = torch.randn(yr.shape, generator=g)
        yr.backward(go)
        xp = x.contiguous(memory_format=torch.channels_last).to(dev).requires_grad_(True)
        wp = w.to(dev).requires_grad_(True)
        yp = HF.depthwise_conv2d(xp, wp, stride, R // 2)
        yp.backward(go.to(dev))
        case = (N, H, W, C, R, stride)
        assert yp.shape == yr.shape, case
        assert (yp.detach().cpu() - yr.detach()).abs().max() <= 1e-5 * max(1.0, yr.abs().max().item()), case
        assert (xp.grad.cpu() - xr.grad).abs().max() <= 1e-5 * max(1.0, xr.grad.abs().max().item()), case
        assert (wp.grad.cpu() - wr.grad).abs().max() <= 2e-5 * max(1.0, wr.grad.abs().max().item()), case


def test_depthwise_conv_emulated(emu_lib):
    _run_dw("cpu")


@pytest.mark.gpu
def test_depthwise_conv_gpu(hip_lib):
    _run_dw("cuda")


def _product():
    from oracle import make_golden as MG
    import omni3d_amd.cubercnn.modeling.backbone  # noqa: F401
    from omni3d_amd.cubercnn.modeling.meta_arch import build_backbone
    torch.manual_seed(17)
    return build_backbone(MG.product_cfg(["MODEL.WEIGHTS", "synthetic://random-init"], "cubercnn_mnasnet_FPN.yaml"))


def _reference_wrapper():
    from oracle import ref_harness as H
    H.install()
    import cubercnn.modeling.backbone  # noqa: F401
    from cubercnn.modeling.backbone.mnasnet import build_mnasnet_fpn_backbone as ref_builder
    from oracle.upstream import ShapeSpec
    return ref_builder(H.reference_cfg("cubercnn_mnasnet_FPN.yaml", ["MODEL.WEIGHTS", "synthetic://random-init"]), ShapeSpec(channels=3))


def _restated_wrapper():
    """what the reference file does with torchvision's model, for the GPU box (no reference checkout there)"""
    from oracle import upstream as U

    class Wrapped(U.Backbone):
        def __init__(self):
            super().__init__()
            self.base = U.tv_mnasnet1_0().layers
            self._out_feature_channels = {"p2": 24, "p3": 40, "p4": 96, "p5": 320, "p6": 320}
            self._out_feature_strides = {"p2": 4, "p3": 8, "p4": 16, "p5": 32, "p6": 64}
            self._out_features = ["p2", "p3", "p4", "p5", "p6"]

        def forward(self, x):
            p2 = self.base[0:9](x)
            p3 = self.base[9](p2)
            p4 = self.base[10:12](p3)
            p5 = self.base[12:14](p4)
            return {"p2": p2, "p3": p3, "p4": p4, "p5": p5, "p6": F.max_pool2d(p5, kernel_size=1, stride=2, padding=0)}
    return U.FPN(Wrapped(), ["p2", "p3", "p4", "p5", "p6"], 256)


@needs_ref
def test_mnasnet_fpn_surface():
    prod, ref = _product(), _reference_wrapper()
    assert list(prod.output_shape().keys()) == list(ref.output_shape().keys()) == ["p2", "p3", "p4", "p5", "p6"]
    assert {k: (v.channels, v.stride) for k, v in prod.output_shape().items()} == \
        {k: (v.channels, v.stride) for k, v in ref.output_shape().items()}
    assert list(prod.state_dict().keys()) == list(ref.state_dict().keys())
    assert {k: tuple(v.shape) for k, v in prod.state_dict().items()} == {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    assert sum(p.numel() for p in prod.bottom_up.parameters()) + 1280 * 1000 + 1000 == 4383312      # torchvision's mnasnet1_0
    assert abs(prod.bottom_up.base[1].momentum - 0.0003) < 1e-9


def _run(dev, size, ref):
    prod = _product()
    ref.load_state_dict(prod.state_dict(), strict=True)
    prod = prod.to(dev).train()
    ref.train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, size, size, generator=g)
    x4 = torch.cat([x, torch.zeros(2, 1, size, size)], 1).contiguous(memory_format=torch.channels_last).to(dev)
    po, ro = prod(x4), ref(x)
    sum((v.float() ** 2).mean() for v in po.values()).backward()
    sum((v ** 2).mean() for v in ro.values()).backward()
    for k in ro:
        assert po[k].shape == ro[k].shape, k
        assert (po[k].detach().cpu() - ro[k].detach()).abs().max() <= 5e-4 * max(1.0, ro[k].abs().max().item()), k
    rg = dict(ref.named_parameters())
    # a BatchNorm bias that feeds (through a linear layer) another BatchNorm has an exactly-zero gradient in exact arithmetic:
    # both sides then hold rounding noise, so the absolute floor is tied to the largest gradient of the network
    floor = 1e-5 * max(float(q.grad.norm()) for q in rg.values() if q.grad is not None)
    for n, p in prod.named_parameters():
        if rg[n].grad is None:                  # base.14 / base.15: the classifier conv the wrapper never calls
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        a, b = p.grad.detach().cpu().contiguous(memory_format=torch.contiguous_format), rg[n].grad
        rel = float((a - b).norm() / (b.norm() + 1e-12))
        assert rel <= (1e-2 if "fpn" in n else 1e-1) or float((a - b).norm()) <= floor, (n, rel, float(b.norm()), floor)
    # running statistics follow torchvision's BN momentum
    rb, pb = dict(ref.named_buffers()), dict(prod.named_buffers())
    for n in ("bottom_up.base.1.running_var", "bottom_up.base.9.1.layers.4.running_mean"):
        assert (pb[n].cpu() - rb[n]).abs().max() <= 1e-5 * max(1.0, rb[n].abs().max().item()), n


@needs_ref
@pytest.mark.skipif(os.environ.get("OMNI_SLOW") != "1", reason="minutes under the host emulator; the GPU variant is the gate")
def test_mnasnet_fpn_emulated(emu_lib):
    _run("cpu", 64, _reference_wrapper())


@pytest.mark.gpu
def test_mnasnet_fpn_gpu(hip_lib):
    _run("cuda", 128, _restated_wrapper())
