"""torchvision-style ResNet-18/34/50/101 bottom-up + FPN builder (`build_resnet_from_vision_fpn_backbone`).

Mirrors /root/reference/cubercnn/modeling/backbone/resnet.py (ResNet wrapper :12-65, builder :68-96):
the wrapper lifts conv1/bn1/maxpool/layer1..4 out of `torchvision.models.resnet{18,34}` (module names,
hence state-dict keys, `layerL.B.{conv1,bn1,conv2,bn2,downsample.0,downsample.1}`), emits p2..p5 and
p6 = max_pool2d(p5, k=1, s=2), and wraps it in FPN(top_block=LastLevelMaxPool()).  torchvision is not
a dependency here: the BasicBlock topology / initialisation is restated on the HIP kernels (7x7/s2 stem
conv and all 3x3 / 1x1 convs = implicit-GEMM MFMA kernel, BN+ReLU(+residual) fused, 3x3/s2 max-pool =
csrc/pool3.hip).  Depths 50 / 101 use torchvision's Bottleneck (1x1 - 3x3 with the stride - 1x1 x 4, `layerL.B.{conv1..3,bn1..3}`;
BASELINE.json's configs[3] is ResNet-34)."""
import torch
from torch import nn

from .... import functional as HF
from ..layers import BatchNorm2d, Conv2d
from ..registries import BACKBONE_REGISTRY
from .fpn import FPN, Backbone, LastLevelMaxPool


class Downsample(nn.Sequential):
    def __init__(self, cin, cout, stride):
        super().__init__(Conv2d(cin, cout, kernel_size=1, stride=stride, bias=False), BatchNorm2d(cout))

    def forward(self, x):
        return self[1](self[0](x))


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = Conv2d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        HF.fanout(x)        # read by conv1 and by the shortcut: the two gradients are summed inside the consumers' kernels
        identity = x if self.downsample is None else self.downsample(x)
        # (bn1 + ReLU are applied inside conv2's Winograd input transform when it takes that path: functional.bn_relu_conv3x3)
        out = HF.bn_relu_conv3x3(self.conv1(x), self.bn1, self.conv2, want_stats=self.training and torch.is_grad_enabled())
        return self.bn2(out, residual=identity, relu=True)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, kernel_size=1, stride=1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = Conv2d(planes, planes * self.expansion, kernel_size=1, stride=1, bias=False)
        self.bn3 = BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        HF.fanout(x)
        identity = x if self.downsample is None else self.downsample(x)
        out = self.bn1(self.conv1(x), relu=True)
        out = self.bn2(self.conv2(out), relu=True)
        return self.bn3(self.conv3(out), residual=identity, relu=True)


class TorchvisionResNet(nn.Module):
    def __init__(self, layers, block=BasicBlock):
        super().__init__()
        self.block = block
        self.inplanes = 64
        self.conv1 = Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0], 1)
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride):
        block, downsample = self.block, None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = Downsample(self.inplanes, planes * block.expansion, stride)
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)


_DEPTHS = {18: ([2, 2, 2, 2], BasicBlock), 34: ([3, 4, 6, 3], BasicBlock), 50: ([3, 4, 6, 3], Bottleneck), 101: ([3, 4, 23, 3], Bottleneck)}


class ResNet(Backbone):
    def __init__(self, cfg, input_shape, pretrained=True):
        super().__init__()
        depth = cfg.MODEL.RESNETS.DEPTH
        if depth not in _DEPTHS:
            raise ValueError("No configuration currently supporting depth of {}".format(depth))
        if pretrained:
            raise RuntimeError("ImageNet ResNet weights are downloaded by the reference via torchvision (resnet.py:16-20); there is "
                               "no network here -- set MODEL.WEIGHTS / MODEL.WEIGHTS_PRETRAIN or load a state dict")
        layers, block = _DEPTHS[depth]
        base = TorchvisionResNet(layers, block)
        e = block.expansion
        self._out_feature_channels = {"p2": 64 * e, "p3": 128 * e, "p4": 256 * e, "p5": 512 * e, "p6": 512 * e}
        for name in ("conv1", "bn1", "relu", "maxpool", "layer1", "layer2", "layer3", "layer4"):
            setattr(self, name, getattr(base, name))
        self._out_feature_strides = {"p2": 4, "p3": 8, "p4": 16, "p5": 32, "p6": 64}
        self._out_features = ["p2", "p3", "p4", "p5", "p6"]

    stage_cut = None      # solver/graphed.py GraphedPipelined: backward is cut between the stages (x -> detached copy of x)
    stage_cut_at = ("p2", "p3")

    def backward_stages(self):
        """{module name: backward stage} for the cut points of `stage_cut_at` (see DLA.backward_stages)"""
        order = [("conv1", None), ("bn1", None), ("layer1", "p2"), ("layer2", "p3"), ("layer3", "p4"), ("layer4", "p5")]
        n_cuts = sum(1 for _, c in order if c in self.stage_cut_at)
        out, seen = {}, 0
        for name, c in order:
            out[name] = 1 + n_cuts - seen
            if c in self.stage_cut_at:
                seen += 1
        return out

    def forward(self, x):
        w = self.conv1.weight
        if x.shape[1] != w.shape[1]:   # 3-channel stem weight against the 4-channel padded image
            w = HF.pad_input_channels(w, x.shape[1], self.__dict__.setdefault("_w_pad", {}))
        x = self.bn1(HF.conv2d(x, w, None, 2, 3, False, self.bn1.training and torch.is_grad_enabled()), relu=True)
        x = HF.max_pool3s2(x)
        on = self.stage_cut is not None and self.training and torch.is_grad_enabled()
        cut = lambda name, t: self.stage_cut(t) if (on and name in self.stage_cut_at) else t      # noqa: E731
        p2 = cut("p2", self.layer1(x))
        p3 = cut("p3", self.layer2(p2))
        p4 = cut("p4", self.layer3(p3))
        p5 = cut("p5", self.layer4(p4))
        return {"p2": p2, "p3": p3, "p4": p4, "p5": p5, "p6": HF.subsample2(p5)}


@BACKBONE_REGISTRY.register()
def build_resnet_from_vision_fpn_backbone(cfg, input_shape, priors=None):
    imagenet_pretrain = cfg.MODEL.WEIGHTS_PRETRAIN + cfg.MODEL.WEIGHTS == ""
    if not cfg.MODEL.RESNETS.TORCHVISION:
        raise NotImplementedError("MODEL.RESNETS.TORCHVISION False (detectron2 MSRA ResNet) is outside the MI355X hot path")
    bottom_up = ResNet(cfg, input_shape, pretrained=imagenet_pretrain)
    return FPN(bottom_up=bottom_up, in_features=cfg.MODEL.FPN.IN_FEATURES, out_channels=cfg.MODEL.FPN.OUT_CHANNELS,
               norm=cfg.MODEL.FPN.NORM, top_block=LastLevelMaxPool(), fuse_type=cfg.MODEL.FPN.FUSE_TYPE)
