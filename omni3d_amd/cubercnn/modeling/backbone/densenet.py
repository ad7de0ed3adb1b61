"""DenseNet-121 bottom-up + FPN builder (`build_densenet_fpn_backbone`, configs/cubercnn_densenet_FPN.yaml).

Mirrors /root/reference/cubercnn/modeling/backbone/densenet.py: the wrapper takes `torchvision.models.densenet121().features`
as `base` (:14-17) and emits p2 = base[0:5] (stem + dense block 1, 256 ch, stride 4), p3 = base[5:7] (transition 1 + block 2, 512),
p4 = base[7:9] (1024), p5 = base[9:] (transition 3 + block 4 + norm5, 1024, no ReLU) and p6 = max_pool2d(p5, k=1, s=2) (:26-36);
the builder wraps it in an FPN WITHOUT a top block (:53-59).  torchvision is not a dependency here: DenseNet-121's topology
(growth 32, blocks (6, 12, 24, 16), bottleneck 4 x 32, 64 stem features; layer = BN-ReLU-1x1 conv-BN-ReLU-3x3 conv on the
concatenation of everything before it; transition = BN-ReLU-1x1 conv (half the channels)-2x2 average pool) and its initialisation
(kaiming_normal_ convs, BN weight 1 / bias 0) are restated on the HIP kernels with torchvision's module names, so state-dict keys
are `base.denseblockB.denselayerL.{norm1,conv1,norm2,conv2}`, `base.transitionT.{norm,conv}`, `base.{conv0,norm0,norm5}`.

Kernels: 7x7/s2 stem + every 1x1 / 3x3 convolution = the implicit-GEMM MFMA kernels, BN + ReLU fused (statistics over the
concatenated features, as in torch), 3x3/s2 max-pool = csrc/pool3.hip, 2x2 average pool = csrc/bn_pool.hip."""
from collections import OrderedDict

import torch
from torch import nn

from .... import functional as HF
from ..layers import BatchNorm2d, Conv2d
from ..registries import BACKBONE_REGISTRY
from .fpn import FPN, Backbone


class _DenseLayer(nn.Module):
    def __init__(self, cin, growth, bn_size):
        super().__init__()
        self.norm1 = BatchNorm2d(cin)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv1 = Conv2d(cin, bn_size * growth, kernel_size=1, stride=1, bias=False)
        self.norm2 = BatchNorm2d(bn_size * growth)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv2 = Conv2d(bn_size * growth, growth, kernel_size=3, stride=1, padding=1, bias=False)

    def forward(self, features):
        x = torch.cat(features, 1) if len(features) > 1 else features[0]
        return self.conv2(self.norm2(self.conv1(self.norm1(x, relu=True)), relu=True))


class _DenseBlock(nn.ModuleDict):
    def __init__(self, num_layers, cin, bn_size, growth):
        super().__init__()
        for i in range(num_layers):
            self["denselayer%d" % (i + 1)] = _DenseLayer(cin + i * growth, growth, bn_size)

    def forward(self, x):
        features = [x]
        for layer in self.values():
            features.append(layer(features))
        return torch.cat(features, 1)


class _Transition(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.norm = BatchNorm2d(cin)
        self.relu = nn.ReLU(inplace=True)
        self.conv = Conv2d(cin, cout, kernel_size=1, stride=1, bias=False)
        self.pool = nn.AvgPool2d(kernel_size=2, stride=2)

    def forward(self, x):
        return HF.avg_pool2(self.conv(self.norm(x, relu=True)))


def densenet121_features(growth=32, block_config=(6, 12, 24, 16), init_features=64, bn_size=4):
    layers = OrderedDict([("conv0", Conv2d(3, init_features, kernel_size=7, stride=2, padding=3, bias=False)),
                          ("norm0", BatchNorm2d(init_features)), ("relu0", nn.ReLU(inplace=True)),
                          ("pool0", nn.MaxPool2d(kernel_size=3, stride=2, padding=1))])
    c = init_features
    for i, n in enumerate(block_config):
        layers["denseblock%d" % (i + 1)] = _DenseBlock(n, c, bn_size, growth)
        c += n * growth
        if i != len(block_config) - 1:
            layers["transition%d" % (i + 1)] = _Transition(c, c // 2)
            c //= 2
    layers["norm5"] = BatchNorm2d(c)
    base = nn.Sequential(layers)
    for m in base.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight)
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)
    return base


class DenseNetBackbone(Backbone):
    def __init__(self, cfg, input_shape, pretrained=True):
        super().__init__()
        if pretrained:
            raise RuntimeError("ImageNet DenseNet weights are downloaded by the reference via torchvision (densenet.py:14); there is "
                               "no network here -- set MODEL.WEIGHTS / MODEL.WEIGHTS_PRETRAIN or load a state dict")
        self.base = densenet121_features()
        self._out_feature_channels = {"p2": 256, "p3": 512, "p4": 1024, "p5": 1024, "p6": 1024}
        self._out_feature_strides = {"p2": 4, "p3": 8, "p4": 16, "p5": 32, "p6": 64}
        self._out_features = ["p2", "p3", "p4", "p5", "p6"]

    def forward(self, x):
        b = self.base
        w = b.conv0.weight
        if x.shape[1] != w.shape[1]:   # 3-channel stem weight against the 4-channel padded image
            w = torch.cat([w, w.new_zeros(w.shape[0], x.shape[1] - w.shape[1], w.shape[2], w.shape[3])], dim=1)
        x = b.norm0(HF.conv2d(x, w, None, 2, 3, False, b.norm0.training and torch.is_grad_enabled()), relu=True)
        db1 = b.denseblock1(HF.max_pool3s2(x))
        db2 = b.denseblock2(b.transition1(db1))
        db3 = b.denseblock3(b.transition2(db2))
        p5 = b.norm5(b.denseblock4(b.transition3(db3)))
        return {"p2": db1, "p3": db2, "p4": db3, "p5": p5, "p6": HF.subsample2(p5)}


@BACKBONE_REGISTRY.register()
def build_densenet_fpn_backbone(cfg, input_shape, priors=None):
    imagenet_pretrain = cfg.MODEL.WEIGHTS_PRETRAIN + cfg.MODEL.WEIGHTS == ""
    bottom_up = DenseNetBackbone(cfg, input_shape, pretrained=imagenet_pretrain)
    return FPN(bottom_up=bottom_up, in_features=cfg.MODEL.FPN.IN_FEATURES, out_channels=cfg.MODEL.FPN.OUT_CHANNELS,
               norm=cfg.MODEL.FPN.NORM, fuse_type=cfg.MODEL.FPN.FUSE_TYPE)
