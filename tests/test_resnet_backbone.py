"""ResNet-34 + FPN(top_block=LastLevelMaxPool) bottom-up (BASELINE configs[3], cubercnn_ResNet34_FPN) against the
oracle's restatement of torchvision resnet34 under the reference wrapper (cubercnn/modeling/backbone/resnet.py)."""
import os

import pytest
import torch


def _build():
    from oracle import make_golden as MG
    from oracle import model_oracle as MO
    from oracle import upstream as U
    import omni3d_amd.cubercnn.modeling.backbone  # noqa: F401
    from omni3d_amd.cubercnn.modeling.meta_arch import build_backbone
    cfg = MG.product_cfg([], "cubercnn_ResNet34_FPN.yaml")
    torch.manual_seed(11)
    prod = build_backbone(cfg)
    ref = U.FPN(MO.ResNet34(), ["p2", "p3", "p4", "p5", "p6"], 256, top_block=U.LastLevelMaxPool())
    ref.load_state_dict(prod.state_dict(), strict=True)
    return prod, ref


def test_resnet_fpn_surface():
    prod, ref = _build()
    assert list(prod.output_shape().keys()) == list(ref.output_shape().keys()) == ["p2", "p3", "p4", "p5", "p6", "p7"]
    assert prod.size_divisibility == ref.size_divisibility == 64
    assert {k: (v.channels, v.stride) for k, v in prod.output_shape().items()} == \
        {k: (v.channels, v.stride) for k, v in ref.output_shape().items()}
    keys = list(prod.state_dict().keys())
    assert "bottom_up.layer2.0.downsample.0.weight" in keys and "bottom_up.layer4.2.bn2.running_var" in keys


def _run(dev, size):
    prod, ref = _build()
    prod = prod.to(dev).train()
    ref.train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 3, size, size, generator=g)
    x4 = torch.cat([x, torch.zeros(1, 1, size, size)], 1).contiguous(memory_format=torch.channels_last).to(dev)
    po, ro = prod(x4), ref(x)
    loss_p = sum((v.float() ** 2).mean() for k, v in po.items() if k != "p7")
    loss_r = sum((v ** 2).mean() for k, v in ro.items() if k != "p7")
    loss_p.backward()
    loss_r.backward()
    for k in ro:
        assert po[k].shape == ro[k].shape, k
        assert (po[k].detach().cpu() - ro[k].detach()).abs().max() <= 2e-4 * max(1.0, ro[k].abs().max().item()), k
    rg = dict(ref.named_parameters())
    for n, p in prod.named_parameters():
        a, b = p.grad.detach().cpu().contiguous(memory_format=torch.contiguous_format), rg[n].grad
        tol = 1e-2 if "fpn" in n else 1e-1
        assert (a - b).norm() <= tol * b.norm() + 1e-7, (n, float((a - b).norm()), float(b.norm()))


@pytest.mark.skipif(os.environ.get("OMNI_SLOW") != "1", reason="minutes under the host emulator; the GPU variant is the gate")
def test_resnet_fpn_emulated(emu_lib):
    _run("cpu", 64)


@pytest.mark.gpu
def test_resnet_fpn_gpu(hip_lib):
    _run("cuda", 128)
