"""DLA bottom-up + FPN builder (`build_dla_from_vision_fpn_backbone`).

Same topology, module names (=> state-dict keys of SURVEY.md Appendix C) and initialisation as
/root/reference/cubercnn/modeling/backbone/dla.py (BasicBlock :40-68, Bottleneck :71-109, Root :156-174, Tree :177-230,
DLA :233-297, variants :312-414, DLABackbone :417-482, builder :484-507): dla34 (BASELINE.json), the Bottleneck
variants dla46_c, dla60, dla102, dla169 and the BottleneckX (grouped 3x3) variants dla46x_c, dla60x_c, dla60x, dla102x,
dla102x2 of MODEL.DLA.TYPE (the 2-channel groups of the *_c ones are zero-padded to the kernels' 4-channel lanes).
Every conv is the implicit-GEMM MFMA kernel, every BN(+ReLU)(+residual) one fused HBM-bound kernel pair; the image
enters as NHWC with C padded 3 -> 4."""
import math
import os

import torch
from torch import nn

from .... import functional as HF
from ....kernels import conv as kconv
from ..layers import BatchNorm2d, Conv2d, GroupedConv2d
from ..registries import BACKBONE_REGISTRY
from .fpn import FPN, Backbone

_ROOT_MULTI_SRC = os.environ.get("OMNI_ROOT_MULTI_SRC", "1") != "0"     # A/B knob: 0 = Root concatenates its children (rounds 1-4)
_SHARE_POOL = os.environ.get("OMNI_DLA_SHARE_POOL", "1") != "0"      # A/B knob: nested trees pool their common input once
_SIDE_STATS = os.environ.get("OMNI_DLA_SIDE_STATS", "1") != "0"      # A/B knob: statistics-only projections of nested trees on the weight-gradient stream


class ConvBNReLU(nn.Sequential):
    """nn.Sequential(conv, bn, relu) with the reference's child names '0','1','2'."""

    def __init__(self, cin, cout, k, stride, pad):
        super().__init__(Conv2d(cin, cout, kernel_size=k, stride=stride, padding=pad, bias=False), BatchNorm2d(cout),
                         nn.ReLU(inplace=True))

    def forward(self, x):
        conv, bn = self[0], self[1]
        w = conv.weight
        if x.shape[1] != w.shape[1]:   # 3-channel stem weight against the 4-channel padded image
            from ....kernels import conv as KC
            if (conv.stride[0] == 1 and conv.padding[0] == 3 and KC.stem_first_eligible(x.shape, w) and bn.training and torch.is_grad_enabled()
                    and not x.requires_grad):
                # round 6: the stem kernels read the filter / write its gradient in the model's 3-channel layout (no padded copy, no
                # slice + add of the gradient at the very end of the step's critical path)
                return bn(HF.stem_first_conv(x, w, True), relu=True)
            w = HF.pad_input_channels(w, x.shape[1], self.__dict__.setdefault("_w_pad", {}))
        return bn(HF.conv2d(x, w, None, conv.stride[0], conv.padding[0], False, bn.training and torch.is_grad_enabled()), relu=True)


class BasicBlock(nn.Module):
    def __init__(self, inplanes, planes, stride=1, dilation=1):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = Conv2d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.stride = stride

    def forward(self, x, residual=None):
        if residual is None:
            residual = x
        # (bn1 + ReLU are applied inside conv2's Winograd input transform when it takes that path: functional.bn_relu_conv3x3)
        out = HF.bn_relu_conv3x3(self.conv1(x), self.bn1, self.conv2, want_stats=self.training and torch.is_grad_enabled())
        return self.bn2(out, residual=residual, relu=True)


class Bottleneck(nn.Module):
    """dla.py:71-109: 1x1 (planes / 2) - 3x3 (stride) - 1x1, BN after each, the Tree's residual added before the last ReLU"""
    expansion = 2

    def __init__(self, inplanes, planes, stride=1, dilation=1):
        super().__init__()
        bottle = planes // Bottleneck.expansion
        self.conv1 = Conv2d(inplanes, bottle, kernel_size=1, bias=False)
        self.bn1 = BatchNorm2d(bottle)
        self.conv2 = Conv2d(bottle, bottle, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = BatchNorm2d(bottle)
        self.conv3 = Conv2d(bottle, planes, kernel_size=1, bias=False)
        self.bn3 = BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.stride = stride

    def forward(self, x, residual=None):
        if residual is None:
            residual = x
        out = self.bn1(self.conv1(x), relu=True)
        out = self.bn2(self.conv2(out), relu=True)
        return self.bn3(self.conv3(out), residual=residual, relu=True)


def bottleneck_x(cardinality):
    """dla.py:112-153: Bottleneck whose 3x3 is grouped (`cardinality` groups) over planes * cardinality / 32 channels"""

    class BottleneckX(nn.Module):
        expansion = 2

        def __init__(self, inplanes, planes, stride=1, dilation=1):
            super().__init__()
            bottle = planes * cardinality // 32
            self.conv1 = Conv2d(inplanes, bottle, kernel_size=1, bias=False)
            self.bn1 = BatchNorm2d(bottle)
            self.conv2 = GroupedConv2d(bottle, bottle, 3, stride=stride, padding=1, groups=cardinality)
            self.bn2 = BatchNorm2d(bottle)
            self.conv3 = Conv2d(bottle, planes, kernel_size=1, bias=False)
            self.bn3 = BatchNorm2d(planes)
            self.relu = nn.ReLU(inplace=True)
            self.stride = stride

        def forward(self, x, residual=None):
            if residual is None:
                residual = x
            out = self.bn1(self.conv1(x), relu=True)
            out = self.bn2(self.conv2(out), relu=True)
            return self.bn3(self.conv3(out), residual=residual, relu=True)
    return BottleneckX


class Root(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, residual):
        super().__init__()
        self.conv = Conv2d(in_channels, out_channels, 1, stride=1, bias=False, padding=(kernel_size - 1) // 2)
        self.bn = BatchNorm2d(out_channels)
        self.relu = nn.ReLU(inplace=True)
        self.residual = residual

    def forward(self, *x):
        conv = self.conv
        if _ROOT_MULTI_SRC and conv.kernel_size == (1, 1) and conv.bias is None and kconv.multi_src_eligible(x, conv.weight):
            # the 1 x 1 convolution reads its reduction slabs from the children directly: torch.cat(x, 1) is never formed (six
            # copies of 7-16 us on the critical path of the 4 x 512 x 512 step, and as many in an inference pass)
            want_stats = self.training and torch.is_grad_enabled()
            y = HF.cat_conv1x1(x, conv.weight, want_stats)
        else:
            # (the concatenation hands each child its slice of the gradient through the child's fan-in slot, see functional.fanout)
            y = conv(HF.cat_channels(x) if (self.training and torch.is_grad_enabled()) else torch.cat(x, 1))
        return self.bn(y, residual=x[0] if self.residual else None, relu=True)


class Project(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__(Conv2d(cin, cout, kernel_size=1, stride=1, bias=False), BatchNorm2d(cout))

    def forward(self, x):
        return self[1](self[0](x))


class Tree(nn.Module):
    def __init__(self, levels, block, in_channels, out_channels, stride=1, level_root=False, root_dim=0,
                 root_kernel_size=1, dilation=1, root_residual=False):
        super().__init__()
        if root_dim == 0:
            root_dim = 2 * out_channels
        if level_root:
            root_dim += in_channels
        if levels == 1:
            self.tree1 = block(in_channels, out_channels, stride, dilation=dilation)
            self.tree2 = block(out_channels, out_channels, 1, dilation=dilation)
            self.root = Root(root_dim, out_channels, root_kernel_size, root_residual)
        else:
            self.tree1 = Tree(levels - 1, block, in_channels, out_channels, stride, root_dim=0,
                              root_kernel_size=root_kernel_size, dilation=dilation, root_residual=root_residual)
            self.tree2 = Tree(levels - 1, block, out_channels, out_channels, root_dim=root_dim + out_channels,
                              root_kernel_size=root_kernel_size, dilation=dilation, root_residual=root_residual)
        self.level_root = level_root
        self.root_dim = root_dim
        self.levels = levels
        self.downsample = nn.MaxPool2d(stride, stride=stride) if stride > 1 else None
        self.project = Project(in_channels, out_channels) if in_channels != out_channels else None

    def forward(self, x, residual=None, children=None, bottom=None):
        children = [] if children is None else children
        # x, bottom and x1 below each have several consumers (max-pool + first block | projection, Root child, nested subtree |
        # next block, its residual, Root child): their gradients are summed inside the consumers' backward kernels
        HF.fanout(x)
        if bottom is None:      # (a nested first subtree pools the same x with the same stride: the outer level passes its result)
            bottom = HF.max_pool2(x) if self.downsample is not None else x
        HF.fanout(bottom)
        if self.levels > 1 and self.project is not None:
            # a nested Tree recomputes `residual` from its own projection and drops this one (dla.py:208-213 of the reference):
            # run it for its BatchNorm's running statistics only -- no autograd graph, no saved activations
            if _SIDE_STATS and HF.side_mode() == "collect":
                # a step being captured: nothing in the step reads that BatchNorm's running statistics, so the 1 x 1 convolution + statistics
                # leave the critical path and replay in the stage's weight-gradient graph (late round 6; OMNI_DLA_SIDE_STATS=0: inline)
                src = bottom.detach()

                def stats_only(src=src):
                    with torch.no_grad():
                        self.project(src)
                HF._side_run(stats_only, (src,))
            else:
                with torch.no_grad():
                    self.project(bottom.detach())
            residual = None
        else:
            residual = self.project(bottom) if self.project is not None else bottom
        if self.level_root:
            children.append(bottom)
        if self.levels == 1:
            x1 = self.tree1(x, residual)
        else:
            t1 = self.tree1
            shared = bottom if _SHARE_POOL and (t1.downsample is None) == (self.downsample is None) else None
            x1 = t1(x, residual, bottom=shared)
        HF.fanout(x1)
        if self.levels == 1:
            x2 = self.tree2(x1)
            return self.root(x2, x1, *children)
        children.append(x1)
        return self.tree2(x1, children=children)


class DLA(nn.Module):
    def __init__(self, levels, channels, block=BasicBlock, residual_root=False):
        super().__init__()
        self.channels = channels
        self.base_layer = ConvBNReLU(3, channels[0], 7, 1, 3)
        self.level0 = ConvBNReLU(channels[0], channels[0], 3, 1, 1)
        self.level1 = ConvBNReLU(channels[0], channels[1], 3, 2, 1)
        self.level2 = Tree(levels[2], block, channels[1], channels[2], 2, level_root=False, root_residual=residual_root)
        self.level3 = Tree(levels[3], block, channels[2], channels[3], 2, level_root=True, root_residual=residual_root)
        self.level4 = Tree(levels[4], block, channels[3], channels[4], 2, level_root=True, root_residual=residual_root)
        self.level5 = Tree(levels[5], block, channels[4], channels[5], 2, level_root=True, root_residual=residual_root)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()


# MODEL.DLA.TYPE -> (levels, channels, block, residual_root, output channels p2..p6)   (dla.py:312-414, 421-450)
_WIDE = {"p2": 128, "p3": 256, "p4": 512, "p5": 1024, "p6": 1024}
DLA_VARIANTS = {
    "dla34": ([1, 1, 1, 2, 2, 1], [16, 32, 64, 128, 256, 512], BasicBlock, False, {"p2": 64, "p3": 128, "p4": 256, "p5": 512, "p6": 512}),
    "dla46_c": ([1, 1, 1, 2, 2, 1], [16, 32, 64, 64, 128, 256], Bottleneck, False, {"p2": 64, "p3": 64, "p4": 128, "p5": 256, "p6": 256}),
    "dla60": ([1, 1, 1, 2, 3, 1], [16, 32, 128, 256, 512, 1024], Bottleneck, False, _WIDE),
    "dla102": ([1, 1, 1, 3, 4, 1], [16, 32, 128, 256, 512, 1024], Bottleneck, True, _WIDE),
    "dla169": ([1, 1, 2, 3, 5, 1], [16, 32, 128, 256, 512, 1024], Bottleneck, True, _WIDE),
    "dla60x": ([1, 1, 1, 2, 3, 1], [16, 32, 128, 256, 512, 1024], bottleneck_x(32), False, _WIDE),
    "dla102x": ([1, 1, 1, 3, 4, 1], [16, 32, 128, 256, 512, 1024], bottleneck_x(32), True, _WIDE),
    "dla102x2": ([1, 1, 1, 3, 4, 1], [16, 32, 128, 256, 512, 1024], bottleneck_x(64), True, _WIDE),
    "dla46x_c": ([1, 1, 1, 2, 2, 1], [16, 32, 64, 64, 128, 256], bottleneck_x(32), False, {"p2": 64, "p3": 64, "p4": 128, "p5": 256, "p6": 256}),
    "dla60x_c": ([1, 1, 1, 2, 3, 1], [16, 32, 64, 64, 128, 256], bottleneck_x(32), False, {"p2": 64, "p3": 64, "p4": 128, "p5": 256, "p6": 256}),
}


def build_dla(name, pretrained=False):
    if pretrained:
        raise RuntimeError("ImageNet DLA weights are downloaded by the reference (dla.py:300-309); there is no network "
                           "here -- set MODEL.WEIGHTS / MODEL.WEIGHTS_PRETRAIN or load a state dict")
    levels, channels, block, residual_root, _ = DLA_VARIANTS[name]
    return DLA(levels, channels, block=block, residual_root=residual_root)


def dla34(pretrained=False, tricks=False):
    return build_dla("dla34", pretrained)


class DLABackbone(Backbone):
    def __init__(self, cfg, input_shape, pretrained=True):
        super().__init__()
        kind = cfg.MODEL.DLA.TYPE
        if kind not in DLA_VARIANTS:
            raise ValueError(f"unknown MODEL.DLA.TYPE {kind}")
        base = build_dla(kind, pretrained=pretrained)
        self._out_feature_channels = dict(DLA_VARIANTS[kind][4])
        for name in ("base_layer", "level0", "level1", "level2", "level3", "level4", "level5"):
            setattr(self, name, getattr(base, name))
        self._out_feature_strides = {"p2": 4, "p3": 8, "p4": 16, "p5": 32, "p6": 64}
        self._out_features = ["p2", "p3", "p4", "p5", "p6"]

    fwd_split = None      # solver/graphed.py GraphedPipelined: callable run once between level 1 and level 2 while a step is captured
    stage_cut = None      # solver/graphed.py GraphedPipelined: backward is cut between the levels (x -> detached copy of x)
    # a cut at level k needs the cuts at all lower levels (see GraphedPipelined).  Measured optimum; "stem" (round 3): the main stream
    # used to wait 0.39 ms for the last weight-gradient graph after its own last kernel, 0.20 ms with the first layer as its own stage
    # (device timestamps of OMNI_PIPE_TIMING=1, profiles/r03_pipe_timing.log): 12.14 -> 12.06 ms / step
    # "l1" (round 4, between level 1 and level 2): the weight gradients of level 2 start while level 1 / 0 are still in backward; the
    # last weight-gradient graph shrinks to the two full-resolution layers: 11.37 -> 11.30 ms (profiles/r04_ab_cut_l1.log)
    stage_cut_at = ("stem", "l1", "p2", "p3")

    def backward_stages(self):
        """{module name: backward stage its parameters' gradients complete in} for the cut points of `stage_cut_at` (stage 0 = the
        heads, 1 = FPN + everything above the topmost cut, counting up towards the input); read by solver/build.py to lay the
        gradient bucket out stage by stage"""
        order = [("base_layer", "stem"), ("level0", None), ("level1", "l1"), ("level2", "p2"), ("level3", "p3"), ("level4", "p4"), ("level5", "p5")]
        n_cuts = sum(1 for _, c in order if c in self.stage_cut_at)
        out, seen = {}, 0
        for name, c in order:
            out[name] = 1 + n_cuts - seen       # modules before the first cut are the LAST stage
            if c in self.stage_cut_at:
                seen += 1
        return out

    def forward(self, x):
        on = self.stage_cut is not None and self.training and torch.is_grad_enabled()
        cut = lambda name, t: self.stage_cut(t) if (on and name in self.stage_cut_at) else t      # noqa: E731
        # ("stem": the first layer's backward -- a BatchNorm backward and a 0.33 ms weight gradient, nothing below it -- as a stage of
        # its own, so the weight gradients of level 2 .. level 0 run beside it instead of after it)
        x = cut("l1", self.level1(self.level0(cut("stem", self.base_layer(x)))))
        if self.fwd_split is not None:       # solver/graphed.py: the forward graph is cut here (no Winograd layer above this point)
            self.fwd_split()
        p2 = cut("p2", self.level2(x))
        p3 = cut("p3", self.level3(p2))
        p4 = cut("p4", self.level4(p3))
        p5 = HF.fanout(cut("p5", self.level5(p4)))        # (FPN lateral + the p6 subsampling; p2..p4 are marked by the next Tree)
        return {"p2": p2, "p3": p3, "p4": p4, "p5": p5, "p6": HF.subsample2(p5)}


@BACKBONE_REGISTRY.register()
def build_dla_from_vision_fpn_backbone(cfg, input_shape, priors=None):
    imagenet_pretrain = cfg.MODEL.WEIGHTS_PRETRAIN + cfg.MODEL.WEIGHTS == ""
    bottom_up = DLABackbone(cfg, input_shape, pretrained=imagenet_pretrain)
    return FPN(bottom_up=bottom_up, in_features=cfg.MODEL.FPN.IN_FEATURES, out_channels=cfg.MODEL.FPN.OUT_CHANNELS,
               norm=cfg.MODEL.FPN.NORM, fuse_type=cfg.MODEL.FPN.FUSE_TYPE)
