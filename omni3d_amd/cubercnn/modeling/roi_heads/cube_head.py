"""`CubeHead` (own ROI_CUBE_HEAD_REGISTRY), default configuration of the reference
(/root/reference/cubercnn/modeling/roi_heads/cube_head.py:19-197): shared fc1/fc2 + ReLU
(`feature_generator.fc1/fc2`), five linear heads `bbox_3D_dims` (3K), `bbox_3D_center_deltas` (2K),
`bbox_3D_pose` (6K), `bbox_3D_center_depth` (K), `bbox_3D_uncertainty` (K, bias 5).

The reference evaluates rotation_6d_to_matrix for ALL K classes of every ROI and gathers the GT class
afterwards (cube_head.py:176, roi_heads.py:447-457).  Here the five heads are ONE fused GEMM
(1024 -> 13K, padded to a multiple of 16) whose raw output goes to the fused decode / loss kernel,
which gathers the class first."""
import torch
from torch import nn

from .... import functional as HF
from ....d2.registry import Registry
from ..layers import FlattenLinear, Linear

ROI_CUBE_HEAD_REGISTRY = Registry("ROI_CUBE_HEAD")


class _FeatureGenerator(nn.Module):
    def __init__(self, channels, size, fc_dim):
        super().__init__()
        self.fc1 = FlattenLinear(channels, size, fc_dim)
        self.fc2 = Linear(fc_dim, fc_dim)
        nn.init.kaiming_uniform_(self.fc2.weight, a=1)
        nn.init.constant_(self.fc2.bias, 0)

    def forward(self, x):
        return self.fc2(self.fc1(x, relu=True), relu=True)


@ROI_CUBE_HEAD_REGISTRY.register()
class CubeHead(nn.Module):
    def __init__(self, cfg, input_shape):
        super().__init__()
        c = cfg.MODEL.ROI_CUBE_HEAD
        self.num_classes = cfg.MODEL.ROI_HEADS.NUM_CLASSES
        if not (c.SHARED_FC and c.Z_TYPE == "direct" and c.POSE_TYPE == "6d" and c.CLUSTER_BINS == 1 and c.NUM_CONV == 0
                and c.NUM_FC == 2 and c.USE_CONFIDENCE):
            if c.POSE_TYPE not in ("6d", "quaternion", "euler"):
                raise ValueError("Cuboid pose type {} is not recognized".format(c.POSE_TYPE))
            raise NotImplementedError("MI355X hot path implements the Base.yaml cube head (shared FC, z direct, 6d pose, confidence)")
        K = self.num_classes
        self.feature_generator = _FeatureGenerator(input_shape.channels, input_shape.height, c.FC_DIM)
        self.bbox_3D_dims = Linear(c.FC_DIM, K * 3)
        self.bbox_3D_center_deltas = Linear(c.FC_DIM, K * 2)
        self.bbox_3D_pose = Linear(c.FC_DIM, K * 6)
        self.bbox_3D_center_depth = Linear(c.FC_DIM, K)
        self.bbox_3D_uncertainty = Linear(c.FC_DIM, K)
        for m in (self.bbox_3D_dims, self.bbox_3D_center_deltas, self.bbox_3D_pose, self.bbox_3D_center_depth,
                  self.bbox_3D_uncertainty):
            nn.init.normal_(m.weight, std=0.001)
            nn.init.constant_(m.bias, 0)
        nn.init.constant_(self.bbox_3D_uncertainty.bias, 5)
        self.fused_dim = (13 * K + 15) // 16 * 16

    def fused_parameters(self):
        """(13K_pad, 1024) weight and bias in the column order of the fused decode kernel:
        [center_deltas 2K | center_depth K | dims 3K | pose 6K | uncertainty K | zero pad]."""
        parts = (self.bbox_3D_center_deltas, self.bbox_3D_center_depth, self.bbox_3D_dims, self.bbox_3D_pose,
                 self.bbox_3D_uncertainty)
        pad = self.fused_dim - 13 * self.num_classes
        w = torch.cat([m.weight for m in parts] + [parts[0].weight.new_zeros(pad, parts[0].weight.shape[1])], dim=0)
        b = torch.cat([m.bias for m in parts] + [parts[0].bias.new_zeros(pad)], dim=0)
        return w, b

    def forward(self, x):
        """x (n, C, 7, 7) ROI features -> raw fused head outputs (n, 13K_pad)."""
        feats = self.feature_generator(x)
        w, b = self.fused_parameters()
        return HF.linear(feats, w, b)


def build_cube_head(cfg, input_shape):
    return ROI_CUBE_HEAD_REGISTRY.get(cfg.MODEL.ROI_CUBE_HEAD.NAME)(cfg, input_shape)
