"""detectron2 `Backbone` / `FPN` (norm="", fuse_type="sum") on the HIP kernels.  Module names
(`fpn_lateral{2..6}`, `fpn_output{2..6}`, `bottom_up`) match the upstream state dict
(SURVEY.md Appendix A.2 / C); built by the reference at cubercnn/modeling/backbone/dla.py:500-506."""
import math

from torch import nn

from .... import functional as HF
from ....d2.layers import ShapeSpec
from ..layers import Conv2d


class Backbone(nn.Module):
    @property
    def size_divisibility(self):
        return 0

    def output_shape(self):
        return {name: ShapeSpec(channels=self._out_feature_channels[name], stride=self._out_feature_strides[name])
                for name in self._out_features}


class LastLevelMaxPool(nn.Module):
    """detectron2 top block: one extra level = max_pool2d(k=1, s=2) of (the bottom-up) p5
    (used by /root/reference/cubercnn/modeling/backbone/resnet.py:92)."""

    def __init__(self):
        super().__init__()
        self.num_levels = 1
        self.in_feature = "p5"

    def forward(self, x):
        return [HF.subsample2(x)]


class FPN(Backbone):
    def __init__(self, bottom_up, in_features, out_channels, norm="", top_block=None, fuse_type="sum"):
        super().__init__()
        if norm != "" or fuse_type != "sum":
            raise NotImplementedError("only FPN(norm='', fuse_type='sum') is on the MI355X hot path")
        shapes = bottom_up.output_shape()
        strides = [shapes[f].stride for f in in_features]
        self._stages = []
        for f in in_features:
            stage = int(math.log2(shapes[f].stride))
            lateral = Conv2d(shapes[f].channels, out_channels, kernel_size=1, bias=True)
            output = Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=True)
            for m in (lateral, output):
                nn.init.kaiming_uniform_(m.weight, a=1)
                m.weight.data = m.weight.data.contiguous(memory_format=__import__("torch").channels_last)
                nn.init.constant_(m.bias, 0)
            self.add_module(f"fpn_lateral{stage}", lateral)
            self.add_module(f"fpn_output{stage}", output)
            self._stages.append(stage)
        self.in_features = tuple(in_features)
        self.bottom_up = bottom_up
        self._out_feature_strides = {f"p{int(math.log2(s))}": s for s in strides}
        self.top_block = top_block
        if top_block is not None:
            for s in range(stage, stage + top_block.num_levels):
                self._out_feature_strides[f"p{s + 1}"] = 2 ** (s + 1)
        self._out_features = list(self._out_feature_strides.keys())
        self._out_feature_channels = {k: out_channels for k in self._out_features}
        self._size_divisibility = strides[-1]

    @property
    def size_divisibility(self):
        return self._size_divisibility

    def forward(self, x):
        feats = self.bottom_up(x)
        results = {}
        prev = None
        for f, stage in zip(reversed(self.in_features), reversed(self._stages)):
            lat = getattr(self, f"fpn_lateral{stage}")(feats[f])
            prev = lat if prev is None else HF.upsample2_add(lat, prev)
            HF.fanout(prev)       # read by its output convolution and by the next finer level's top-down sum
            results[f"p{stage}"] = HF.fanout(getattr(self, f"fpn_output{stage}")(prev))      # ... by the RPN head and by ROIAlign
        if self.top_block is not None:
            src = feats[self.top_block.in_feature] if self.top_block.in_feature in feats else results[self.top_block.in_feature]
            for i, t in enumerate(self.top_block(src)):
                results[f"p{self._stages[-1] + 1 + i}"] = t
        return {k: results[k] for k in self._out_features}
