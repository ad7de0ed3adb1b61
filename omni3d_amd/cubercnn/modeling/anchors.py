"""detectron2 DefaultAnchorGenerator (configs/Base.yaml:45-47 of the reference): cell anchors
w = sqrt(s^2 / r), h = r * w; grid shifts arange(0, W*stride, stride); order (H, W, A), levels
concatenated p2 -> p6.  Computed once per feature-map shape on the host in the
exact float32 sequence of the upstream code and cached on the device (integer indexing is exact)."""
import math

import torch

from ...d2.config import configurable
from .registries import ANCHOR_GENERATOR_REGISTRY


def _broadcast(params, n):
    if not isinstance(params[0], (list, tuple)):
        return [list(params)] * n
    if len(params) == 1:
        return [list(params[0])] * n
    assert len(params) == n
    return [list(p) for p in params]


@ANCHOR_GENERATOR_REGISTRY.register()
class DefaultAnchorGenerator(torch.nn.Module):
    box_dim = 4

    @configurable
    def __init__(self, *, sizes, aspect_ratios, strides, offset=0.5):
        super().__init__()
        self.strides = list(strides)
        n = len(self.strides)
        self.sizes, self.aspect_ratios = _broadcast(sizes, n), _broadcast(aspect_ratios, n)
        self.offset = offset
        self._cache = {}

    @classmethod
    def from_config(cls, cfg, input_shape):
        return {"sizes": cfg.MODEL.ANCHOR_GENERATOR.SIZES, "aspect_ratios": cfg.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS,
                "strides": [x.stride for x in input_shape], "offset": cfg.MODEL.ANCHOR_GENERATOR.OFFSET}

    @property
    def num_anchors(self):
        return [len(s) * len(a) for s, a in zip(self.sizes, self.aspect_ratios)]

    def cell_anchors(self, level):
        out = []
        for size in self.sizes[level]:
            area = size ** 2.0
            for r in self.aspect_ratios[level]:
                w = math.sqrt(area / r)
                h = r * w
                out.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
        return torch.tensor(out)

    def grid(self, hw_list, device):
        """-> (A_total, 4) anchors for feature maps of sizes hw_list, cached per (shapes, device)."""
        key = (tuple(hw_list), str(device))
        if key not in self._cache:
            per_level = []
            for level, ((H, W), stride) in enumerate(zip(hw_list, self.strides)):
                sx = torch.arange(self.offset * stride, W * stride, step=stride, dtype=torch.float32)
                sy = torch.arange(self.offset * stride, H * stride, step=stride, dtype=torch.float32)
                yy, xx = torch.meshgrid(sy, sx, indexing="ij")
                shifts = torch.stack((xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)), dim=1)
                per_level.append((shifts.view(-1, 1, 4) + self.cell_anchors(level).view(1, -1, 4)).reshape(-1, 4))
            self._cache[key] = torch.cat(per_level).contiguous().to(device)
        return self._cache[key]


def build_anchor_generator(cfg, input_shape):
    return ANCHOR_GENERATOR_REGISTRY.get(cfg.MODEL.ANCHOR_GENERATOR.NAME)(cfg, input_shape)
