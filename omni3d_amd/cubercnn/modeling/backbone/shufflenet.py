"""ShuffleNet-V2 x1.0 bottom-up + FPN builder (`build_shufflenet_fpn_backbone`, configs/cubercnn_shufflenet_FPN.yaml).

Mirrors /root/reference/cubercnn/modeling/backbone/shufflenet.py: the wrapper lifts conv1 / maxpool / stage2..4 / conv5 out of
`torchvision.models.shufflenet_v2_x1_0()` (:14-20) and emits p2 = maxpool(conv1(x)) (24 ch, stride 4), p3 = stage2 (116),
p4 = stage3 (232), p5 = stage4 (464) and p6 = max_pool2d(p5, k=1, s=2) (:29-43); conv5 stays in the state dict unused, as in the
reference.  FPN without a top block (:60-66).

torchvision is not a dependency here: ShuffleNet-V2's topology -- 3x3/s2 stem conv 3->24 + BN + ReLU, 3x3/s2 max-pool, stages of
[4, 8, 4] units with [116, 232, 464] channels; a stride-2 unit runs both halves on the whole input (depthwise 3x3/s2 + BN + 1x1 + BN +
ReLU | 1x1 + BN + ReLU + depthwise 3x3/s2 + BN + 1x1 + BN + ReLU), a stride-1 unit keeps the first half and runs the second branch
on the other; concat, then channel shuffle with 2 groups -- is restated with torchvision's module names and PyTorch's default
initialisation (2 278 604 parameters with the classifier).

Kernels: the same 1x1 / depthwise / BatchNorm / pooling kernels as the other backbones.  Stage 2 works on 58-channel halves: every
kernel here moves 4 channels per lane, so those tensors (and the matching weight / BatchNorm vectors) are zero-padded to 60
channels around each kernel call and sliced back; a padded channel stays exactly zero through conv, BatchNorm (zero gamma / beta)
and their gradients.  chunk / concat / shuffle are index permutations and stay torch views + copies."""
import torch
import torch.nn.functional as F
from torch import nn

from .... import functional as HF
from ....kernels import bnpool
from ..layers import BatchNorm2d, Conv2d, DepthwiseConv2d
from ..registries import BACKBONE_REGISTRY
from .fpn import FPN, Backbone


def _up4(c):
    return (c + 3) // 4 * 4


def _pad_channels(x, c4):
    return x if x.shape[1] == c4 else F.pad(x, (0, 0, 0, 0, 0, c4 - x.shape[1]))


def _pad_vec(v, c4, value=0.0):
    return v if v.shape[0] == c4 else F.pad(v, (0, c4 - v.shape[0]), value=value)


def _conv1x1(x, m):
    """m: Conv2d(cin, cout, 1, bias=False) on a tensor whose channel counts may not be multiples of 4"""
    cin, cout = m.weight.shape[1], m.weight.shape[0]
    if cin % 4 == 0 and cout % 4 == 0:
        return m(x)
    w = F.pad(m.weight, (0, 0, 0, 0, 0, _up4(cin) - cin, 0, _up4(cout) - cout))
    return HF.conv2d(_pad_channels(x, _up4(cin)), w)[:, :cout]


def _depthwise(x, m):
    c = m.weight.shape[0]
    if c % 4 == 0:
        return m(x)
    w = F.pad(m.weight, (0, 0, 0, 0, 0, 0, 0, _up4(c) - c))
    return HF.depthwise_conv2d(_pad_channels(x, _up4(c)), w, m.stride[0], m.padding[0])[:, :c]


def _bn(x, m, relu=False):
    c = m.num_features
    if c % 4 == 0:
        return m(x, relu=relu)
    c4 = _up4(c)
    xp = _pad_channels(x, c4)
    gamma, beta = _pad_vec(m.weight, c4), _pad_vec(m.bias, c4)          # zero gamma / beta: the pad channels stay exactly zero
    if m.training:
        mean, var = _pad_vec(m.running_mean.detach(), c4).clone(), _pad_vec(m.running_var.detach(), c4, 1.0).clone()
        y = HF.batch_norm_train(xp, gamma, beta, mean, var, None, relu, m.eps, m.momentum)
        with torch.no_grad():
            m.running_mean.copy_(mean[:c])
            m.running_var.copy_(var[:c])
            if not m.defer_counter:
                m.num_batches_tracked += 1
        return y[:, :c]
    scale = gamma * torch.rsqrt(_pad_vec(m.running_var, c4, 1.0) + m.eps)
    scale_shift = torch.cat([scale, beta - _pad_vec(m.running_mean, c4) * scale]).detach().contiguous()
    return bnpool.bn_apply(xp.contiguous(memory_format=torch.channels_last), scale_shift, None, relu)[:, :c]


def channel_shuffle(x, groups):
    n, c, h, w = x.shape
    return x.reshape(n, groups, c // groups, h, w).transpose(1, 2).reshape(n, c, h, w)


class InvertedResidual(nn.Module):
    def __init__(self, inp, oup, stride):
        super().__init__()
        self.stride = stride
        bf = oup // 2
        assert stride != 1 or inp == bf << 1
        if stride > 1:
            self.branch1 = nn.Sequential(DepthwiseConv2d(inp, 3, stride=stride, padding=1), BatchNorm2d(inp),
                                         Conv2d(inp, bf, kernel_size=1, stride=1, padding=0, bias=False), BatchNorm2d(bf), nn.ReLU(inplace=True))
        else:
            self.branch1 = nn.Sequential()
        self.branch2 = nn.Sequential(Conv2d(inp if stride > 1 else bf, bf, kernel_size=1, stride=1, padding=0, bias=False), BatchNorm2d(bf),
                                     nn.ReLU(inplace=True), DepthwiseConv2d(bf, 3, stride=stride, padding=1), BatchNorm2d(bf),
                                     Conv2d(bf, bf, kernel_size=1, stride=1, padding=0, bias=False), BatchNorm2d(bf), nn.ReLU(inplace=True))

    def _branch1(self, x):
        b = self.branch1
        return _bn(_conv1x1(_bn(_depthwise(x, b[0]), b[1]), b[2]), b[3], relu=True)

    def _branch2(self, x):
        b = self.branch2
        y = _bn(_conv1x1(x, b[0]), b[1], relu=True)
        y = _bn(_depthwise(y, b[3]), b[4])
        return _bn(_conv1x1(y, b[5]), b[6], relu=True)

    def forward(self, x):
        if self.stride == 1:
            x1, x2 = x.chunk(2, dim=1)
            out = torch.cat((x1, self._branch2(x2)), dim=1)
        else:
            out = torch.cat((self._branch1(x), self._branch2(x)), dim=1)
        return channel_shuffle(out, 2)


class ShufflenetBackbone(Backbone):
    def __init__(self, cfg, input_shape, pretrained=True):
        super().__init__()
        if pretrained:
            raise RuntimeError("ImageNet ShuffleNet weights are downloaded by the reference via torchvision (shufflenet.py:14); there is "
                               "no network here -- set MODEL.WEIGHTS / MODEL.WEIGHTS_PRETRAIN or load a state dict")
        repeats, channels = [4, 8, 4], [24, 116, 232, 464, 1024]
        self.conv1 = nn.Sequential(Conv2d(3, channels[0], kernel_size=3, stride=2, padding=1, bias=False), BatchNorm2d(channels[0]),
                                   nn.ReLU(inplace=True))
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        cin = channels[0]
        for name, rep, cout in zip(("stage2", "stage3", "stage4"), repeats, channels[1:]):
            setattr(self, name, nn.Sequential(InvertedResidual(cin, cout, 2), *[InvertedResidual(cout, cout, 1) for _ in range(rep - 1)]))
            cin = cout
        self.conv5 = nn.Sequential(Conv2d(cin, channels[-1], kernel_size=1, stride=1, padding=0, bias=False), BatchNorm2d(channels[-1]),
                                   nn.ReLU(inplace=True))
        self._out_feature_channels = {"p2": 24, "p3": 116, "p4": 232, "p5": 464, "p6": 464}
        self._out_feature_strides = {"p2": 4, "p3": 8, "p4": 16, "p5": 32, "p6": 64}
        self._out_features = ["p2", "p3", "p4", "p5", "p6"]

    def forward(self, x):
        w = self.conv1[0].weight
        if x.shape[1] != w.shape[1]:   # 3-channel stem weight against the 4-channel padded image
            w = torch.cat([w, w.new_zeros(w.shape[0], x.shape[1] - w.shape[1], w.shape[2], w.shape[3])], dim=1)
        bn = self.conv1[1]
        x = bn(HF.conv2d(x, w, None, 2, 1, False, bn.training and torch.is_grad_enabled()), relu=True)
        p2 = HF.max_pool3s2(x)
        p3 = self.stage2(p2)
        p4 = self.stage3(p3)
        p5 = self.stage4(p4)
        return {"p2": p2, "p3": p3, "p4": p4, "p5": p5, "p6": HF.subsample2(p5)}


@BACKBONE_REGISTRY.register()
def build_shufflenet_fpn_backbone(cfg, input_shape, priors=None):
    imagenet_pretrain = cfg.MODEL.WEIGHTS_PRETRAIN + cfg.MODEL.WEIGHTS == ""
    bottom_up = ShufflenetBackbone(cfg, input_shape, pretrained=imagenet_pretrain)
    return FPN(bottom_up=bottom_up, in_features=cfg.MODEL.FPN.IN_FEATURES, out_channels=cfg.MODEL.FPN.OUT_CHANNELS,
               norm=cfg.MODEL.FPN.NORM, fuse_type=cfg.MODEL.FPN.FUSE_TYPE)
