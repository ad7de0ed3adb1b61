"""DenseNet-121 + FPN bottom-up (configs/cubercnn_densenet_FPN.yaml) against the REFERENCE's own wrapper
(cubercnn/modeling/backbone/densenet.py, run under oracle/ref_harness.py) over the oracle's restatement of torchvision's
densenet121 (oracle/upstream.py; torchvision itself is not vendored, so that restatement is unpinned)."""
import os

import pytest
import torch

REF = "/root/reference"


def _build():
    from oracle import make_golden as MG
    from oracle import ref_harness as H
    import omni3d_amd.cubercnn.modeling.backbone  # noqa: F401
    from omni3d_amd.cubercnn.modeling.meta_arch import build_backbone
    ov = ["MODEL.WEIGHTS", "synthetic://random-init"]
    torch.manual_seed(13)
    prod = build_backbone(MG.product_cfg(ov, "cubercnn_densenet_FPN.yaml"))
    H.install()
    import cubercnn.modeling.backbone  # noqa: F401   (registers the reference's builders)
    from cubercnn.modeling.backbone.densenet import build_densenet_fpn_backbone as ref_builder
    from oracle.upstream import ShapeSpec
    ref = ref_builder(H.reference_cfg("cubercnn_densenet_FPN.yaml", ov), ShapeSpec(channels=3))
    ref.load_state_dict(prod.state_dict(), strict=True)
    return prod, ref


needs_ref = pytest.mark.skipif(not os.path.exists(REF), reason="needs the reference checkout (build container only)")


@needs_ref
def test_densenet_fpn_surface():
    prod, ref = _build()
    assert list(prod.output_shape().keys()) == list(ref.output_shape().keys()) == ["p2", "p3", "p4", "p5", "p6"]
    assert prod.size_divisibility == ref.size_divisibility == 64
    assert {k: (v.channels, v.stride) for k, v in prod.output_shape().items()} == \
        {k: (v.channels, v.stride) for k, v in ref.output_shape().items()}
    keys = list(prod.state_dict().keys())
    assert keys == list(ref.state_dict().keys())
    assert "bottom_up.base.denseblock3.denselayer24.conv2.weight" in keys and "bottom_up.base.transition2.norm.running_var" in keys
    assert sum(p.numel() for p in prod.bottom_up.parameters()) == 6953856      # densenet121.features


def _run(dev, size):
    prod, ref = _build()
    prod = prod.to(dev).train()
    ref.train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, size, size, generator=g)
    x4 = torch.cat([x, torch.zeros(2, 1, size, size)], 1).contiguous(memory_format=torch.channels_last).to(dev)
    po, ro = prod(x4), ref(x)
    loss_p = sum((v.float() ** 2).mean() for v in po.values())
    loss_r = sum((v ** 2).mean() for v in ro.values())
    loss_p.backward()
    loss_r.backward()
    for k in ro:
        assert po[k].shape == ro[k].shape, k
        assert (po[k].detach().cpu() - ro[k].detach()).abs().max() <= 5e-4 * max(1.0, ro[k].abs().max().item()), k
    rg = dict(ref.named_parameters())
    worst = 0.0
    for n, p in prod.named_parameters():
        a, b = p.grad.detach().cpu().contiguous(memory_format=torch.contiguous_format), rg[n].grad
        rel = float((a - b).norm() / (b.norm() + 1e-12))
        worst = max(worst, rel)
        assert rel <= (1e-2 if "fpn" in n else 1e-1) or float((a - b).norm()) <= 1e-6, (n, rel, float(b.norm()))
    return worst


@needs_ref
@pytest.mark.skipif(os.environ.get("OMNI_SLOW") != "1", reason="minutes under the host emulator; the GPU variant is the gate")
def test_densenet_fpn_emulated(emu_lib):
    _run("cpu", 64)


@pytest.mark.gpu
def test_densenet_fpn_gpu(hip_lib):
    """on the GPU box the reference checkout is absent: the same comparison against the oracle restatement wrapped the way the
    reference file wraps it (slices of `features`, p6 = stride-2 subsample, FPN without a top block)"""
    from oracle import make_golden as MG
    from oracle import upstream as U
    import omni3d_amd.cubercnn.modeling.backbone  # noqa: F401
    from omni3d_amd.cubercnn.modeling.meta_arch import build_backbone
    import torch.nn.functional as F

    class Wrapped(U.Backbone):
        def __init__(self):
            super().__init__()
            self.base = U.tv_densenet121().features
            self._out_feature_channels = {"p2": 256, "p3": 512, "p4": 1024, "p5": 1024, "p6": 1024}
            self._out_feature_strides = {"p2": 4, "p3": 8, "p4": 16, "p5": 32, "p6": 64}
            self._out_features = ["p2", "p3", "p4", "p5", "p6"]

        def forward(self, x):
            db1 = self.base[0:5](x)
            db2 = self.base[5:7](db1)
            db3 = self.base[7:9](db2)
            p5 = self.base[9:](db3)
            return {"p2": db1, "p3": db2, "p4": db3, "p5": p5, "p6": F.max_pool2d(p5, kernel_size=1, stride=2, padding=0)}

    torch.manual_seed(13)
    prod = build_backbone(MG.product_cfg(["MODEL.WEIGHTS", "synthetic://random-init"], "cubercnn_densenet_FPN.yaml")).cuda().train()
    ref = U.FPN(Wrapped(), ["p2", "p3", "p4", "p5", "p6"], 256)
    ref.load_state_dict({k: v.cpu() for k, v in prod.state_dict().items()}, strict=True)
    ref.train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 128, 128, generator=g)
    x4 = torch.cat([x, torch.zeros(2, 1, 128, 128)], 1).contiguous(memory_format=torch.channels_last).cuda()
    po, ro = prod(x4), ref(x)
    sum((v.float() ** 2).mean() for v in po.values()).backward()
    sum((v ** 2).mean() for v in ro.values()).backward()
    for k in ro:
        assert (po[k].detach().cpu() - ro[k].detach()).abs().max() <= 5e-4 * max(1.0, ro[k].abs().max().item()), k
    rg = dict(ref.named_parameters())
    for n, p in prod.named_parameters():
        a, b = p.grad.detach().cpu().contiguous(memory_format=torch.contiguous_format), rg[n].grad
        rel = float((a - b).norm() / (b.norm() + 1e-12))
        assert rel <= (1e-2 if "fpn" in n else 1e-1) or float((a - b).norm()) <= 1e-6, (n, rel)
