"""`CubeHead` (own ROI_CUBE_HEAD_REGISTRY) of the reference (/root/reference/cubercnn/modeling/roi_heads/cube_head.py:19-197):
FC feature generator(s) + ReLU -- one shared `feature_generator.fc1..fcN` (SHARED_FC, configs/Base.yaml) or one per output
group (`feature_generator_{XY,dims,pose,Z,conf}`) -- and the linear heads `bbox_3D_dims` (3K), `bbox_3D_center_deltas` (2K),
`bbox_3D_pose` (6K / 4K / 3K for POSE_TYPE 6d / quaternion / euler), `bbox_3D_center_depth` (K x max(CLUSTER_BINS, 1), bin-major)
and, with USE_CONFIDENCE, `bbox_3D_uncertainty` (K, bias 5).  Parameter names equal the reference's, so its checkpoints load.

The reference evaluates the rotation for ALL K classes of every ROI and gathers the GT class afterwards (cube_head.py:175-185,
roi_heads.py:447-457).  Here the heads are ONE fused GEMM (fc_dim -> width*K, padded to a multiple of 16) whose raw output goes
to the fused decode / loss kernel (csrc/cube_head.hip), which gathers the class (and, with CLUSTER_BINS > 1, the depth cluster)
first and applies the pose / depth / dimension parameterisation named by the ROI heads' `cube_mode`."""
import torch
from torch import nn

from .... import functional as HF
from ....d2.registry import Registry
from ....kernels import det
from ..layers import FlattenLinear, Linear

ROI_CUBE_HEAD_REGISTRY = Registry("ROI_CUBE_HEAD")


class _FeatureGenerator(nn.Module):
    """fc1 (on the flattened ROI feature) .. fcN, each followed by ReLU (cube_head.py:63-103)"""

    def __init__(self, channels, size, fc_dim, num_fc=2):
        super().__init__()
        self.num_fc = num_fc
        self.fc1 = FlattenLinear(channels, size, fc_dim)
        for k in range(2, num_fc + 1):
            fc = Linear(fc_dim, fc_dim)
            nn.init.kaiming_uniform_(fc.weight, a=1)
            nn.init.constant_(fc.bias, 0)
            setattr(self, f"fc{k}", fc)

    def forward(self, x):
        x = self.fc1(x, relu=True)
        for k in range(2, self.num_fc + 1):
            x = getattr(self, f"fc{k}")(x, relu=True)
        return x


@ROI_CUBE_HEAD_REGISTRY.register()
class CubeHead(nn.Module):
    def __init__(self, cfg, input_shape):
        super().__init__()
        c = cfg.MODEL.ROI_CUBE_HEAD
        self.num_classes = K = cfg.MODEL.ROI_HEADS.NUM_CLASSES
        if c.POSE_TYPE not in det.POSE_WIDTH:
            raise ValueError("Cuboid pose type {} is not recognized".format(c.POSE_TYPE))
        self.cluster_bins = bins = c.CLUSTER_BINS if c.CLUSTER_BINS > 1 else 1                 # cube_head.py:112
        if c.NUM_FC < 1:
            raise NotImplementedError("CubeHead needs NUM_FC >= 1 (the reference builds no conv layers either, cube_head.py:49-103)")
        self.use_conf, self.shared_fc, self.pose_type = bool(c.USE_CONFIDENCE), bool(c.SHARED_FC), c.POSE_TYPE
        gen = lambda: _FeatureGenerator(input_shape.channels, input_shape.height, c.FC_DIM, c.NUM_FC)   # noqa: E731
        if self.shared_fc:
            self.feature_generator = gen()
        else:
            self.feature_generator_XY, self.feature_generator_dims = gen(), gen()
            self.feature_generator_pose, self.feature_generator_Z = gen(), gen()
            if self.use_conf:
                self.feature_generator_conf = gen()
        self.bbox_3D_dims = Linear(c.FC_DIM, K * 3)
        self.bbox_3D_center_deltas = Linear(c.FC_DIM, K * 2)
        self.bbox_3D_pose = Linear(c.FC_DIM, K * det.POSE_WIDTH[c.POSE_TYPE])
        self.bbox_3D_center_depth = Linear(c.FC_DIM, K * bins)                                # cube_head.py:136, viewed (n, bins, K)
        heads = [self.bbox_3D_dims, self.bbox_3D_center_deltas, self.bbox_3D_pose, self.bbox_3D_center_depth]
        if self.use_conf:
            self.bbox_3D_uncertainty = Linear(c.FC_DIM, K)
            heads.append(self.bbox_3D_uncertainty)
        for m in heads:
            nn.init.normal_(m.weight, std=0.001)
            nn.init.constant_(m.bias, 0)
        if self.use_conf:
            nn.init.constant_(self.bbox_3D_uncertainty.bias, 5)
        self.width = 5 + bins + det.POSE_WIDTH[c.POSE_TYPE] + int(self.use_conf)      # columns per class of the fused output
        self.fused_dim = (self.width * K + 15) // 16 * 16

    def _parts(self):
        """the linear heads in the column order of the fused decode kernel: [center_deltas 2K | center_depth K*bins | dims 3K | pose | uncertainty K]"""
        parts = [self.bbox_3D_center_deltas, self.bbox_3D_center_depth, self.bbox_3D_dims, self.bbox_3D_pose]
        return parts + ([self.bbox_3D_uncertainty] if self.use_conf else [])

    def fused_param_groups(self):
        """read by solver/build.py tag_fused_groups (shared FC only: the heads are then ONE GEMM): the optimizer keeps these back to
        back (+ zero padding) in its buckets, so the fused matrix and its gradient are views"""
        if not self.shared_fc:
            return []
        parts = self._parts()
        pad = self.fused_dim - self.width * self.num_classes
        return [([m.weight for m in parts], pad * parts[0].weight.shape[1]), ([m.bias for m in parts], pad)]

    def fused_parameters(self):
        """(width*K padded, fc_dim) weight and bias of the one GEMM that replaces the separate heads (shared FC only)."""
        parts = self._parts()
        pad = self.fused_dim - self.width * self.num_classes
        w = torch.cat([m.weight for m in parts] + [parts[0].weight.new_zeros(pad, parts[0].weight.shape[1])], dim=0)
        b = torch.cat([m.bias for m in parts] + [parts[0].bias.new_zeros(pad)], dim=0)
        return w, b

    def forward(self, x):
        """x (n, C, 7, 7) ROI features -> raw fused head outputs (n, width*K padded)."""
        if self.shared_fc:
            parts = self._parts()
            return HF.fused_linear(self.feature_generator(x), [m.weight for m in parts], [m.bias for m in parts], self.fused_dim)
        # SHARED_FC False (cube_head.py:165-173): every output group has its own FC stack
        gens = [self.feature_generator_XY, self.feature_generator_Z, self.feature_generator_dims, self.feature_generator_pose]
        if self.use_conf:
            gens.append(self.feature_generator_conf)
        outs = []
        for g, m in zip(gens, self._parts()):
            n = m.weight.shape[0]
            p4 = (n + 15) // 16 * 16 - n              # the GEMM kernels want an output width that is a multiple of 4
            w = torch.cat([m.weight, m.weight.new_zeros(p4, m.weight.shape[1])], 0) if p4 else m.weight
            b = torch.cat([m.bias, m.bias.new_zeros(p4)], 0) if p4 else m.bias
            outs.append(HF.linear(g(x), w, b)[:, :n])
        pad = self.fused_dim - self.width * self.num_classes
        if pad:
            outs.append(outs[0].new_zeros(outs[0].shape[0], pad))
        return torch.cat(outs, dim=1).contiguous()


def build_cube_head(cfg, input_shape):
    return ROI_CUBE_HEAD_REGISTRY.get(cfg.MODEL.ROI_CUBE_HEAD.NAME)(cfg, input_shape)
