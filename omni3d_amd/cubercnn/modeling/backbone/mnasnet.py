"""MNASNet-1.0 bottom-up + FPN builder (`build_mnasnet_fpn_backbone`, configs/cubercnn_mnasnet_FPN.yaml).

Mirrors /root/reference/cubercnn/modeling/backbone/mnasnet.py: the wrapper takes `torchvision.models.mnasnet1_0().layers` as
`base` (:14-17) and emits p2 = base[0:9] (stem + first stack, 24 ch, stride 4), p3 = base[9] (40), p4 = base[10:12] (96),
p5 = base[12:14] (320) and p6 = max_pool2d(p5, k=1, s=2) (:26-39); base[14:17] (the 1280-wide classifier conv) stays in the state
dict unused, as in the reference.  The builder wraps it in an FPN without a top block (:55-61).

torchvision is not a dependency here: MNASNet-1.0's topology -- 3x3/s2 stem conv 3->32, depthwise 3x3 + 1x1 to 16, then stacks of
inverted residuals (1x1 expand, k x k depthwise, 1x1 project; BN after each, ReLU after the first two, identity skip when the
shapes allow): (24, k3, s2, x3 expansion, 3 blocks), (40, k5, s2, x3, 3), (80, k5, s2, x6, 3), (96, k3, s1, x6, 2),
(192, k5, s2, x6, 4), (320, k3, s1, x6, 1) -- BatchNorm momentum 1 - 0.9997 and kaiming_normal_(fan_out) initialisation are
restated with torchvision's module layout (`base.8.0.layers.3.weight` ...), 3 102 312 parameters.

Kernels: 1x1 and the stem convolution = implicit-GEMM MFMA kernels, depthwise k x k = csrc/depthwise.hip (HBM-bound), BN + ReLU
(+ skip) fused."""
import torch
from torch import nn

from .... import functional as HF
from ..layers import BatchNorm2d, Conv2d, DepthwiseConv2d
from ..registries import BACKBONE_REGISTRY
from .fpn import FPN, Backbone

_BN_MOMENTUM = 1 - 0.9997


class _InvertedResidual(nn.Module):
    def __init__(self, cin, cout, kernel_size, stride, expansion):
        super().__init__()
        mid = cin * expansion
        self.apply_residual = cin == cout and stride == 1
        self.layers = nn.Sequential(
            Conv2d(cin, mid, kernel_size=1, bias=False), BatchNorm2d(mid, momentum=_BN_MOMENTUM), nn.ReLU(inplace=True),
            DepthwiseConv2d(mid, kernel_size, stride=stride, padding=kernel_size // 2), BatchNorm2d(mid, momentum=_BN_MOMENTUM),
            nn.ReLU(inplace=True),
            Conv2d(mid, cout, kernel_size=1, bias=False), BatchNorm2d(cout, momentum=_BN_MOMENTUM))

    def forward(self, x):
        L = self.layers
        y = L[1](L[0](x), relu=True)
        y = L[4](L[3](y), relu=True)
        return L[7](L[6](y), residual=x if self.apply_residual else None)


def _stack(cin, cout, kernel_size, stride, expansion, repeats):
    return nn.Sequential(_InvertedResidual(cin, cout, kernel_size, stride, expansion),
                         *[_InvertedResidual(cout, cout, kernel_size, 1, expansion) for _ in range(1, repeats)])


def mnasnet1_0_layers():
    d = [32, 16, 24, 40, 80, 96, 192, 320]
    layers = nn.Sequential(
        Conv2d(3, d[0], kernel_size=3, padding=1, stride=2, bias=False), BatchNorm2d(d[0], momentum=_BN_MOMENTUM), nn.ReLU(inplace=True),
        DepthwiseConv2d(d[0], 3, stride=1, padding=1), BatchNorm2d(d[0], momentum=_BN_MOMENTUM), nn.ReLU(inplace=True),
        Conv2d(d[0], d[1], kernel_size=1, padding=0, stride=1, bias=False), BatchNorm2d(d[1], momentum=_BN_MOMENTUM),
        _stack(d[1], d[2], 3, 2, 3, 3), _stack(d[2], d[3], 5, 2, 3, 3), _stack(d[3], d[4], 5, 2, 6, 3), _stack(d[4], d[5], 3, 1, 6, 2),
        _stack(d[5], d[6], 5, 2, 6, 4), _stack(d[6], d[7], 3, 1, 6, 1),
        Conv2d(d[7], 1280, kernel_size=1, padding=0, stride=1, bias=False), BatchNorm2d(1280, momentum=_BN_MOMENTUM), nn.ReLU(inplace=True))
    for m in layers.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            if m.groups == 1:
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)
    return layers


class MNASNetBackbone(Backbone):
    def __init__(self, cfg, input_shape, pretrained=True):
        super().__init__()
        if pretrained:
            raise RuntimeError("ImageNet MNASNet weights are downloaded by the reference via torchvision (mnasnet.py:14); there is "
                               "no network here -- set MODEL.WEIGHTS / MODEL.WEIGHTS_PRETRAIN or load a state dict")
        self.base = mnasnet1_0_layers()
        self._out_feature_channels = {"p2": 24, "p3": 40, "p4": 96, "p5": 320, "p6": 320}
        self._out_feature_strides = {"p2": 4, "p3": 8, "p4": 16, "p5": 32, "p6": 64}
        self._out_features = ["p2", "p3", "p4", "p5", "p6"]

    def forward(self, x):
        b = self.base
        w = b[0].weight
        if x.shape[1] != w.shape[1]:   # 3-channel stem weight against the 4-channel padded image
            w = torch.cat([w, w.new_zeros(w.shape[0], x.shape[1] - w.shape[1], w.shape[2], w.shape[3])], dim=1)
        x = b[1](HF.conv2d(x, w, None, 2, 1, False, b[1].training and torch.is_grad_enabled()), relu=True)
        x = b[4](b[3](x), relu=True)
        x = b[7](b[6](x))
        p2 = b[8](x)
        p3 = b[9](p2)
        p4 = b[11](b[10](p3))
        p5 = b[13](b[12](p4))
        return {"p2": p2, "p3": p3, "p4": p4, "p5": p5, "p6": HF.subsample2(p5)}


@BACKBONE_REGISTRY.register()
def build_mnasnet_fpn_backbone(cfg, input_shape, priors=None):
    imagenet_pretrain = cfg.MODEL.WEIGHTS_PRETRAIN + cfg.MODEL.WEIGHTS == ""
    bottom_up = MNASNetBackbone(cfg, input_shape, pretrained=imagenet_pretrain)
    return FPN(bottom_up=bottom_up, in_features=cfg.MODEL.FPN.IN_FEATURES, out_channels=cfg.MODEL.FPN.OUT_CHANNELS,
               norm=cfg.MODEL.FPN.NORM, fuse_type=cfg.MODEL.FPN.FUSE_TYPE)
