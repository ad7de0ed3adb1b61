"""oracle/upstream.py -- TEST INFRASTRUCTURE (CPU oracle).  Never imported by the product.

Plain-PyTorch (CPU, fp32) restatement of the arithmetic that the reference reaches through its
UN-VENDORED dependencies -- detectron2 (wheel for torch1.8 => v0.5-0.6), torchvision 0.9.1,
pytorch3d (unpinned), fvcore -- none of which is under /root/reference or installed here
(reference README.md:54-71).  Every piece cites the reference call site that uses it.  The
semantics are restated from the published sources of those projects.

Two uses:
  1. `oracle/ref_harness.py` installs these objects under the upstream import paths so the
     reference's OWN hot-path files (/root/reference/cubercnn/...) execute unchanged on CPU in the
     build container; that run generates the golden fixtures in tests/golden/.
  2. The per-op functions (roi_align, nms, Matcher, Box2BoxTransform, anchors, ...) are the
     float/integer oracles the kernel parity tests compare against (they travel to the GPU box).

Containers (Boxes, Instances, ImageList, CfgNode, Registry, EventStorage) are plumbing shared with
the product package omni3d_amd.d2; all arithmetic here is independent of the product kernels.
"""
import inspect
import math
from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from omni3d_amd.d2.config import CfgNode, configurable, get_cfg  # noqa: F401  (plumbing)
from omni3d_amd.d2.events import EventStorage, get_event_storage  # noqa: F401
from omni3d_amd.d2.layers import ShapeSpec, cat, nonzero_tuple  # noqa: F401
from omni3d_amd.d2.registry import Registry
from omni3d_amd.d2.structures import Boxes, BoxMode, ImageList, Instances  # noqa: F401

# ------------------------------------------------------------------------------------------------
# detectron2.structures.boxes: pairwise_iou / pairwise_ioa   (rpn.py:62,100; roi_heads.py:881,892)
# ------------------------------------------------------------------------------------------------


def pairwise_intersection(boxes1: Boxes, boxes2: Boxes) -> torch.Tensor:
    b1, b2 = boxes1.tensor, boxes2.tensor
    wh = torch.min(b1[:, None, 2:], b2[:, 2:]) - torch.max(b1[:, None, :2], b2[:, :2])
    wh.clamp_(min=0)
    return wh.prod(dim=2)


def pairwise_iou(boxes1: Boxes, boxes2: Boxes) -> torch.Tensor:
    area1, area2 = boxes1.area(), boxes2.area()
    inter = pairwise_intersection(boxes1, boxes2)
    return torch.where(inter > 0, inter / (area1[:, None] + area2 - inter), torch.zeros(1, dtype=inter.dtype, device=inter.device))


def pairwise_ioa(boxes1: Boxes, boxes2: Boxes) -> torch.Tensor:
    area2 = boxes2.area()
    inter = pairwise_intersection(boxes1, boxes2)
    return torch.where(inter > 0, inter / area2, torch.zeros(1, dtype=inter.dtype, device=inter.device))


# ------------------------------------------------------------------------------------------------
# torchvision.ops.nms / batched_nms  (via detectron2.layers.batched_nms: fast_rcnn.py:105, RPN)
# ------------------------------------------------------------------------------------------------


def nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """Greedy NMS: descending score; a later box is suppressed iff IoU > thr (strict);
    areas (x2-x1)*(y2-y1); returns kept indices in descending-score order."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    order = torch.sort(scores, descending=True, stable=True)[1]
    b = boxes[order]
    areas = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    n = b.shape[0]
    suppressed = torch.zeros(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        if i + 1 < n:
            xx1 = torch.maximum(b[i, 0], b[i + 1:, 0])
            yy1 = torch.maximum(b[i, 1], b[i + 1:, 1])
            xx2 = torch.minimum(b[i, 2], b[i + 1:, 2])
            yy2 = torch.minimum(b[i, 3], b[i + 1:, 3])
            w = (xx2 - xx1).clamp(min=0)
            h = (yy2 - yy1).clamp(min=0)
            inter = w * h
            iou = inter / (areas[i] + areas[i + 1:] - inter)
            suppressed[i + 1:] |= iou > iou_threshold
    return order[torch.tensor(keep, dtype=torch.int64)]


def batched_nms(boxes, scores, idxs, iou_threshold):
    """Per-group NMS on RAW coordinates (SURVEY.md A.7 spec decision: same selection as the
    torchvision coordinate-offset trick without its rounding).  Output sorted by score."""
    assert boxes.shape[-1] == 4
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    boxes = boxes.float()
    keep_mask = torch.zeros_like(scores, dtype=torch.bool)
    for gid in torch.unique(idxs):
        sel = torch.where(idxs == gid)[0]
        k = nms(boxes[sel], scores[sel], iou_threshold)
        keep_mask[sel[k]] = True
    keep = torch.where(keep_mask)[0]
    return keep[torch.sort(scores[keep], descending=True, stable=True)[1]]


def cross_entropy(input, target, *, reduction="mean", **kwargs):
    if target.numel() == 0 and reduction == "mean":
        return input.sum() * 0.0
    return F.cross_entropy(input, target, reduction=reduction, **kwargs)


# ------------------------------------------------------------------------------------------------
# fvcore.nn  (rpn.py:11,261; fast_rcnn.py:7,220,250; cube_head.py:8,71)
# ------------------------------------------------------------------------------------------------


def smooth_l1_loss(input, target, beta: float, reduction: str = "none"):
    if beta < 1e-5:
        loss = torch.abs(input - target)
    else:
        n = torch.abs(input - target)
        cond = n < beta
        loss = torch.where(cond, 0.5 * n ** 2 / beta, n - 0.5 * beta)
    if reduction == "mean":
        loss = loss.mean() if loss.numel() > 0 else 0.0 * loss.sum()
    elif reduction == "sum":
        loss = loss.sum()
    return loss


def c2_xavier_fill(module):
    nn.init.kaiming_uniform_(module.weight, a=1)
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


def c2_msra_fill(module):
    nn.init.kaiming_normal_(module.weight, mode="fan_out", nonlinearity="relu")
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


# ------------------------------------------------------------------------------------------------
# detectron2.modeling.box_regression.Box2BoxTransform  (rpn.py:15,259; fast_rcnn.py:216,246)
# ------------------------------------------------------------------------------------------------

_DEFAULT_SCALE_CLAMP = math.log(1000.0 / 16)


class Box2BoxTransform:
    def __init__(self, weights, scale_clamp: float = _DEFAULT_SCALE_CLAMP):
        self.weights = weights
        self.scale_clamp = scale_clamp

    def get_deltas(self, src_boxes, target_boxes):
        src_widths = src_boxes[:, 2] - src_boxes[:, 0]
        src_heights = src_boxes[:, 3] - src_boxes[:, 1]
        src_ctr_x = src_boxes[:, 0] + 0.5 * src_widths
        src_ctr_y = src_boxes[:, 1] + 0.5 * src_heights
        target_widths = target_boxes[:, 2] - target_boxes[:, 0]
        target_heights = target_boxes[:, 3] - target_boxes[:, 1]
        target_ctr_x = target_boxes[:, 0] + 0.5 * target_widths
        target_ctr_y = target_boxes[:, 1] + 0.5 * target_heights
        wx, wy, ww, wh = self.weights
        dx = wx * (target_ctr_x - src_ctr_x) / src_widths
        dy = wy * (target_ctr_y - src_ctr_y) / src_heights
        dw = ww * torch.log(target_widths / src_widths)
        dh = wh * torch.log(target_heights / src_heights)
        deltas = torch.stack((dx, dy, dw, dh), dim=1)
        assert (src_widths > 0).all().item(), "Input boxes to Box2BoxTransform are not valid!"
        return deltas

    def apply_deltas(self, deltas, boxes):
        deltas = deltas.float()
        boxes = boxes.to(deltas.dtype)
        widths = boxes[:, 2] - boxes[:, 0]
        heights = boxes[:, 3] - boxes[:, 1]
        ctr_x = boxes[:, 0] + 0.5 * widths
        ctr_y = boxes[:, 1] + 0.5 * heights
        wx, wy, ww, wh = self.weights
        dx = deltas[:, 0::4] / wx
        dy = deltas[:, 1::4] / wy
        dw = deltas[:, 2::4] / ww
        dh = deltas[:, 3::4] / wh
        dw = torch.clamp(dw, max=self.scale_clamp)
        dh = torch.clamp(dh, max=self.scale_clamp)
        pred_ctr_x = dx * widths[:, None] + ctr_x[:, None]
        pred_ctr_y = dy * heights[:, None] + ctr_y[:, None]
        pred_w = torch.exp(dw) * widths[:, None]
        pred_h = torch.exp(dh) * heights[:, None]
        x1 = pred_ctr_x - 0.5 * pred_w
        y1 = pred_ctr_y - 0.5 * pred_h
        x2 = pred_ctr_x + 0.5 * pred_w
        y2 = pred_ctr_y + 0.5 * pred_h
        pred_boxes = torch.stack((x1, y1, x2, y2), dim=-1)
        return pred_boxes.reshape(deltas.shape)


def _dense_box_regression_loss(anchors, box2box_transform, pred_anchor_deltas, gt_boxes, fg_mask,
                               box_reg_loss_type="smooth_l1", smooth_l1_beta=0.0):
    """detectron2 `_dense_box_regression_loss` (used only when OBJECTNESS_UNCERTAINTY == 'none',
    rpn.py:181-190)."""
    anchors = type(anchors[0]).cat(anchors).tensor if isinstance(anchors[0], Boxes) else cat(anchors)
    assert box_reg_loss_type == "smooth_l1"
    gt_anchor_deltas = torch.stack([box2box_transform.get_deltas(anchors, k) for k in gt_boxes])
    return smooth_l1_loss(cat(pred_anchor_deltas, dim=1)[fg_mask], gt_anchor_deltas[fg_mask], beta=smooth_l1_beta,
                          reduction="sum")


# ------------------------------------------------------------------------------------------------
# detectron2.modeling.matcher.Matcher  (rpn.py:63 via self.anchor_matcher; roi_heads.py:882)
# ------------------------------------------------------------------------------------------------


class Matcher:
    def __init__(self, thresholds: List[float], labels: List[int], allow_low_quality_matches: bool = False):
        thresholds = thresholds[:]
        assert thresholds[0] > 0
        thresholds.insert(0, -float("inf"))
        thresholds.append(float("inf"))
        assert all(low <= high for (low, high) in zip(thresholds[:-1], thresholds[1:]))
        assert all(l in [-1, 0, 1] for l in labels)
        assert len(labels) == len(thresholds) - 1
        self.thresholds = thresholds
        self.labels = labels
        self.allow_low_quality_matches = allow_low_quality_matches

    def __call__(self, match_quality_matrix):
        assert match_quality_matrix.dim() == 2
        if match_quality_matrix.numel() == 0:
            default_matches = match_quality_matrix.new_full((match_quality_matrix.size(1),), 0, dtype=torch.int64)
            default_match_labels = match_quality_matrix.new_full((match_quality_matrix.size(1),), self.labels[0], dtype=torch.int8)
            return default_matches, default_match_labels
        assert torch.all(match_quality_matrix >= 0)
        matched_vals, matches = match_quality_matrix.max(dim=0)
        match_labels = matches.new_full(matches.size(), 1, dtype=torch.int8)
        for (l, low, high) in zip(self.labels, self.thresholds[:-1], self.thresholds[1:]):
            low_high = (matched_vals >= low) & (matched_vals < high)
            match_labels[low_high] = l
        if self.allow_low_quality_matches:
            self.set_low_quality_matches_(match_labels, match_quality_matrix)
        return matches, match_labels

    def set_low_quality_matches_(self, match_labels, match_quality_matrix):
        highest_quality_foreach_gt, _ = match_quality_matrix.max(dim=1)
        _, pred_inds_with_highest_quality = nonzero_tuple(match_quality_matrix == highest_quality_foreach_gt[:, None])
        match_labels[pred_inds_with_highest_quality] = 1


# ------------------------------------------------------------------------------------------------
# detectron2.modeling.anchor_generator.DefaultAnchorGenerator  (Base.yaml:45-47)
# ------------------------------------------------------------------------------------------------

ANCHOR_GENERATOR_REGISTRY = Registry("ANCHOR_GENERATOR")


def _broadcast_params(params, num_features, name):
    assert isinstance(params, (list, tuple)) and len(params)
    if not isinstance(params[0], (list, tuple)):
        return [params] * num_features
    if len(params) == 1:
        return list(params) * num_features
    assert len(params) == num_features
    return params


def _create_grid_offsets(size, stride, offset, device):
    grid_height, grid_width = size
    shifts_x = torch.arange(offset * stride, grid_width * stride, step=stride, dtype=torch.float32, device=device)
    shifts_y = torch.arange(offset * stride, grid_height * stride, step=stride, dtype=torch.float32, device=device)
    shift_y, shift_x = torch.meshgrid(shifts_y, shifts_x, indexing="ij")
    return shift_x.reshape(-1), shift_y.reshape(-1)


@ANCHOR_GENERATOR_REGISTRY.register()
class DefaultAnchorGenerator(nn.Module):
    box_dim = 4

    @configurable
    def __init__(self, *, sizes, aspect_ratios, strides, offset=0.5):
        super().__init__()
        self.strides = strides
        self.num_features = len(self.strides)
        sizes = _broadcast_params(sizes, self.num_features, "sizes")
        aspect_ratios = _broadcast_params(aspect_ratios, self.num_features, "aspect_ratios")
        self.cell_anchors = [self.generate_cell_anchors(s, a).float() for s, a in zip(sizes, aspect_ratios)]
        self.offset = offset
        assert 0.0 <= self.offset < 1.0, self.offset

    @classmethod
    def from_config(cls, cfg, input_shape: List[ShapeSpec]):
        return {"sizes": cfg.MODEL.ANCHOR_GENERATOR.SIZES, "aspect_ratios": cfg.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS,
                "strides": [x.stride for x in input_shape], "offset": cfg.MODEL.ANCHOR_GENERATOR.OFFSET}

    @property
    def num_anchors(self):
        return [len(c) for c in self.cell_anchors]

    @property
    def num_cell_anchors(self):
        return self.num_anchors

    def generate_cell_anchors(self, sizes=(32, 64, 128, 256, 512), aspect_ratios=(0.5, 1, 2)):
        anchors = []
        for size in sizes:
            area = size ** 2.0
            for aspect_ratio in aspect_ratios:
                w = math.sqrt(area / aspect_ratio)
                h = aspect_ratio * w
                anchors.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
        return torch.tensor(anchors)

    def forward(self, features: List[torch.Tensor]):
        grid_sizes = [f.shape[-2:] for f in features]
        out = []
        for size, stride, base in zip(grid_sizes, self.strides, self.cell_anchors):
            shift_x, shift_y = _create_grid_offsets(size, stride, self.offset, base.device)
            shifts = torch.stack((shift_x, shift_y, shift_x, shift_y), dim=1)
            out.append(Boxes((shifts.view(-1, 1, 4) + base.view(1, -1, 4)).reshape(-1, 4)))
        return out


def build_anchor_generator(cfg, input_shape):
    return ANCHOR_GENERATOR_REGISTRY.get(cfg.MODEL.ANCHOR_GENERATOR.NAME)(cfg, input_shape)


# ------------------------------------------------------------------------------------------------
# detectron2.modeling.proposal_generator: RPN base, StandardRPNHead, find_top_rpn_proposals
#   (base class of RPNWithIgnore, rpn.py:16,20; Base.yaml:49-60)
# ------------------------------------------------------------------------------------------------

PROPOSAL_GENERATOR_REGISTRY = Registry("PROPOSAL_GENERATOR")
RPN_HEAD_REGISTRY = Registry("RPN_HEAD")


class _ConvRelu(nn.Conv2d):
    """detectron2.layers.Conv2d with activation=ReLU (state-dict keys `weight`, `bias`)."""

    def forward(self, x):
        return F.relu(super().forward(x))


@RPN_HEAD_REGISTRY.register()
class StandardRPNHead(nn.Module):
    @configurable
    def __init__(self, *, in_channels: int, num_anchors: int, box_dim: int = 4, conv_dims=(-1,)):
        super().__init__()
        assert len(conv_dims) == 1
        out_channels = in_channels if conv_dims[0] == -1 else conv_dims[0]
        self.conv = _ConvRelu(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.objectness_logits = nn.Conv2d(out_channels, num_anchors, kernel_size=1, stride=1)
        self.anchor_deltas = nn.Conv2d(out_channels, num_anchors * box_dim, kernel_size=1, stride=1)
        for layer in self.modules():
            if isinstance(layer, nn.Conv2d):
                nn.init.normal_(layer.weight, std=0.01)
                nn.init.constant_(layer.bias, 0)

    @classmethod
    def from_config(cls, cfg, input_shape):
        in_channels = [s.channels for s in input_shape]
        assert len(set(in_channels)) == 1
        anchor_generator = build_anchor_generator(cfg, input_shape)
        num_anchors = anchor_generator.num_anchors
        assert len(set(num_anchors)) == 1
        return {"in_channels": in_channels[0], "num_anchors": num_anchors[0], "box_dim": anchor_generator.box_dim,
                "conv_dims": cfg.MODEL.RPN.CONV_DIMS}

    def forward(self, features: List[torch.Tensor]):
        logits, deltas = [], []
        for x in features:
            t = self.conv(x)
            logits.append(self.objectness_logits(t))
            deltas.append(self.anchor_deltas(t))
        return logits, deltas


def build_rpn_head(cfg, input_shape):
    return RPN_HEAD_REGISTRY.get(cfg.MODEL.RPN.HEAD_NAME)(cfg, input_shape)


def find_top_rpn_proposals(proposals, pred_objectness_logits, image_sizes, nms_thresh, pre_nms_topk, post_nms_topk,
                           min_box_size, training):
    num_images = len(image_sizes)
    device = proposals[0].device
    topk_scores, topk_proposals, level_ids = [], [], []
    batch_idx = torch.arange(num_images, device=device)
    for level_id, (proposals_i, logits_i) in enumerate(zip(proposals, pred_objectness_logits)):
        Hi_Wi_A = logits_i.shape[1]
        num_proposals_i = min(Hi_Wi_A, pre_nms_topk)
        logits_i, idx = logits_i.sort(descending=True, dim=1, stable=True)
        topk_scores_i = logits_i.narrow(1, 0, num_proposals_i)
        topk_idx = idx.narrow(1, 0, num_proposals_i)
        topk_proposals.append(proposals_i[batch_idx[:, None], topk_idx])
        topk_scores.append(topk_scores_i)
        level_ids.append(torch.full((num_proposals_i,), level_id, dtype=torch.int64, device=device))
    topk_scores = cat(topk_scores, dim=1)
    topk_proposals = cat(topk_proposals, dim=1)
    level_ids = cat(level_ids, dim=0)
    results = []
    for n, image_size in enumerate(image_sizes):
        boxes = Boxes(topk_proposals[n])
        scores_per_img = topk_scores[n]
        lvl = level_ids
        valid_mask = torch.isfinite(boxes.tensor).all(dim=1) & torch.isfinite(scores_per_img)
        if not valid_mask.all():
            if training:
                raise FloatingPointError("Predicted boxes or scores contain Inf/NaN. Training has diverged.")
            boxes, scores_per_img, lvl = boxes[valid_mask], scores_per_img[valid_mask], lvl[valid_mask]
        boxes.clip(image_size)
        keep = boxes.nonempty(threshold=min_box_size)
        if keep.sum().item() != len(boxes):
            boxes, scores_per_img, lvl = boxes[keep], scores_per_img[keep], lvl[keep]
        keep = batched_nms(boxes.tensor, scores_per_img, lvl, nms_thresh)
        keep = keep[:post_nms_topk]
        res = Instances(image_size)
        res.proposal_boxes = boxes[keep]
        res.objectness_logits = scores_per_img[keep]
        results.append(res)
    return results


@PROPOSAL_GENERATOR_REGISTRY.register()
class RPN(nn.Module):
    @configurable
    def __init__(self, *, in_features, head, anchor_generator, anchor_matcher, box2box_transform, batch_size_per_image,
                 positive_fraction, pre_nms_topk, post_nms_topk, nms_thresh=0.7, min_box_size=0.0,
                 anchor_boundary_thresh=-1.0, loss_weight=1.0, box_reg_loss_type="smooth_l1", smooth_l1_beta=0.0):
        super().__init__()
        self.in_features = in_features
        self.rpn_head = head
        self.anchor_generator = anchor_generator
        self.anchor_matcher = anchor_matcher
        self.box2box_transform = box2box_transform
        self.batch_size_per_image = batch_size_per_image
        self.positive_fraction = positive_fraction
        self.pre_nms_topk = {True: pre_nms_topk[0], False: pre_nms_topk[1]}
        self.post_nms_topk = {True: post_nms_topk[0], False: post_nms_topk[1]}
        self.nms_thresh = nms_thresh
        self.min_box_size = float(min_box_size)
        self.anchor_boundary_thresh = anchor_boundary_thresh
        if isinstance(loss_weight, float):
            loss_weight = {"loss_rpn_cls": loss_weight, "loss_rpn_loc": loss_weight}
        self.loss_weight = loss_weight
        self.box_reg_loss_type = box_reg_loss_type
        self.smooth_l1_beta = smooth_l1_beta

    @classmethod
    def from_config(cls, cfg, input_shape: Dict[str, ShapeSpec]):
        in_features = cfg.MODEL.RPN.IN_FEATURES
        ret = {
            "in_features": in_features,
            "min_box_size": cfg.MODEL.PROPOSAL_GENERATOR.MIN_SIZE,
            "nms_thresh": cfg.MODEL.RPN.NMS_THRESH,
            "batch_size_per_image": cfg.MODEL.RPN.BATCH_SIZE_PER_IMAGE,
            "positive_fraction": cfg.MODEL.RPN.POSITIVE_FRACTION,
            "loss_weight": {"loss_rpn_cls": cfg.MODEL.RPN.LOSS_WEIGHT,
                            "loss_rpn_loc": cfg.MODEL.RPN.BBOX_REG_LOSS_WEIGHT * cfg.MODEL.RPN.LOSS_WEIGHT},
            "anchor_boundary_thresh": cfg.MODEL.RPN.BOUNDARY_THRESH,
            "box2box_transform": Box2BoxTransform(weights=cfg.MODEL.RPN.BBOX_REG_WEIGHTS),
            "box_reg_loss_type": cfg.MODEL.RPN.BBOX_REG_LOSS_TYPE,
            "smooth_l1_beta": cfg.MODEL.RPN.SMOOTH_L1_BETA,
        }
        ret["pre_nms_topk"] = (cfg.MODEL.RPN.PRE_NMS_TOPK_TRAIN, cfg.MODEL.RPN.PRE_NMS_TOPK_TEST)
        ret["post_nms_topk"] = (cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN, cfg.MODEL.RPN.POST_NMS_TOPK_TEST)
        ret["anchor_generator"] = build_anchor_generator(cfg, [input_shape[f] for f in in_features])
        ret["anchor_matcher"] = Matcher(cfg.MODEL.RPN.IOU_THRESHOLDS, cfg.MODEL.RPN.IOU_LABELS, allow_low_quality_matches=True)
        ret["head"] = build_rpn_head(cfg, [input_shape[f] for f in in_features])
        return ret

    def forward(self, images, features, gt_instances=None):
        features = [features[f] for f in self.in_features]
        anchors = self.anchor_generator(features)
        pred_objectness_logits, pred_anchor_deltas = self.rpn_head(features)
        pred_objectness_logits = [score.permute(0, 2, 3, 1).flatten(1) for score in pred_objectness_logits]
        pred_anchor_deltas = [
            x.view(x.shape[0], -1, self.anchor_generator.box_dim, x.shape[-2], x.shape[-1]).permute(0, 3, 4, 1, 2).flatten(1, -2)
            for x in pred_anchor_deltas]
        if self.training:
            assert gt_instances is not None, "RPN requires gt_instances in training!"
            gt_labels, gt_boxes = self.label_and_sample_anchors(anchors, gt_instances)
            losses = self.losses(anchors, pred_objectness_logits, gt_labels, pred_anchor_deltas, gt_boxes)
        else:
            losses = {}
        proposals = self.predict_proposals(anchors, pred_objectness_logits, pred_anchor_deltas, images.image_sizes)
        return proposals, losses

    def predict_proposals(self, anchors, pred_objectness_logits, pred_anchor_deltas, image_sizes):
        with torch.no_grad():
            pred_proposals = self._decode_proposals(anchors, pred_anchor_deltas)
            return find_top_rpn_proposals(pred_proposals, pred_objectness_logits, image_sizes, self.nms_thresh,
                                          self.pre_nms_topk[self.training], self.post_nms_topk[self.training],
                                          self.min_box_size, self.training)

    def _decode_proposals(self, anchors, pred_anchor_deltas):
        N = pred_anchor_deltas[0].shape[0]
        proposals = []
        for anchors_i, pred_anchor_deltas_i in zip(anchors, pred_anchor_deltas):
            B = anchors_i.tensor.size(1)
            pred_anchor_deltas_i = pred_anchor_deltas_i.reshape(-1, B)
            anchors_i = anchors_i.tensor.unsqueeze(0).expand(N, -1, -1).reshape(-1, B)
            proposals_i = self.box2box_transform.apply_deltas(pred_anchor_deltas_i, anchors_i)
            proposals.append(proposals_i.view(N, -1, B))
        return proposals


def build_proposal_generator(cfg, input_shape):
    name = cfg.MODEL.PROPOSAL_GENERATOR.NAME
    if name == "PrecomputedProposals":
        return None
    return PROPOSAL_GENERATOR_REGISTRY.get(name)(cfg, input_shape)


def add_ground_truth_to_proposals(gt, proposals):
    assert gt is not None and len(proposals) == len(gt)
    if len(proposals) == 0:
        return proposals
    return [add_ground_truth_to_proposals_single_image(g, p) for g, p in zip(gt, proposals)]


def add_ground_truth_to_proposals_single_image(gt, proposals):
    if isinstance(gt, Boxes):
        gt = Instances(proposals.image_size, gt_boxes=gt)
    gt_boxes = gt.gt_boxes
    device = proposals.objectness_logits.device
    gt_logit_value = math.log((1.0 - 1e-10) / (1 - (1.0 - 1e-10)))
    gt_logits = gt_logit_value * torch.ones(len(gt_boxes), device=device)
    gt_proposal = Instances(proposals.image_size, **gt.get_fields())
    gt_proposal.proposal_boxes = gt_boxes
    gt_proposal.objectness_logits = gt_logits
    for key in proposals.get_fields().keys():
        assert gt_proposal.has(key), f"The attribute '{key}' in `proposals` does not exist in `gt`"
    return Instances.cat([proposals, gt_proposal])


# ------------------------------------------------------------------------------------------------
# torchvision.ops.roi_align (aligned=True, sampling_ratio=0) + detectron2.modeling.poolers.ROIPooler
#   (roi_heads.py:166-171,267,362)
# ------------------------------------------------------------------------------------------------


def roi_align(input, rois, output_size, spatial_scale, sampling_ratio=0, aligned=True):
    """input (N,C,H,W), rois (K,5)=[batch, x1,y1,x2,y2] -> (K,C,P,P).  Differentiable w.r.t. input.
    Follows torchvision's CPU kernel: adaptive grid ceil(roi/P), bilinear taps with the
    y<-1||y>H -> 0 rule, clamp at 0, top-edge snap, mean over max(grid_h*grid_w,1) samples."""
    P = output_size if isinstance(output_size, int) else output_size[0]
    N, C, H, W = input.shape
    K = rois.shape[0]
    out = input.new_zeros((K, C, P, P))
    if K == 0:
        return out
    off = 0.5 if aligned else 0.0
    bidx = rois[:, 0].long()
    sw = rois[:, 1] * spatial_scale - off
    sh = rois[:, 2] * spatial_scale - off
    ew = rois[:, 3] * spatial_scale - off
    eh = rois[:, 4] * spatial_scale - off
    rw, rh = ew - sw, eh - sh
    if not aligned:
        rw, rh = rw.clamp(min=1.0), rh.clamp(min=1.0)
    bin_h, bin_w = rh / P, rw / P
    gh = torch.ceil(rh / P).long() if sampling_ratio <= 0 else torch.full_like(bidx, sampling_ratio)
    gw = torch.ceil(rw / P).long() if sampling_ratio <= 0 else torch.full_like(bidx, sampling_ratio)
    flat = input.permute(0, 2, 3, 1).reshape(N * H * W, C)
    key = gh * 100003 + gw
    for kval in torch.unique(key):
        sel = torch.where(key == kval)[0]
        g_h, g_w = int(gh[sel[0]]), int(gw[sel[0]])
        if g_h <= 0 or g_w <= 0:
            continue
        count = float(max(g_h * g_w, 1))
        ph = torch.arange(P, dtype=input.dtype)
        iy = torch.arange(g_h, dtype=input.dtype)
        ix = torch.arange(g_w, dtype=input.dtype)
        # y[k, ph, iy], x[k, pw, ix]
        y = sh[sel, None, None] + ph[None, :, None] * bin_h[sel, None, None] + (iy[None, None, :] + 0.5) * bin_h[sel, None, None] / g_h
        x = sw[sel, None, None] + ph[None, :, None] * bin_w[sel, None, None] + (ix[None, None, :] + 0.5) * bin_w[sel, None, None] / g_w
        y = y.reshape(len(sel), P * g_h)
        x = x.reshape(len(sel), P * g_w)
        vy = ~((y < -1.0) | (y > H))
        vx = ~((x < -1.0) | (x > W))
        y = y.clamp(min=0)
        x = x.clamp(min=0)
        y_low = y.long()
        x_low = x.long()
        ytop = y_low >= H - 1
        xtop = x_low >= W - 1
        y_low = torch.where(ytop, torch.full_like(y_low, H - 1), y_low)
        x_low = torch.where(xtop, torch.full_like(x_low, W - 1), x_low)
        y_high = torch.where(ytop, y_low, y_low + 1)
        x_high = torch.where(xtop, x_low, x_low + 1)
        y = torch.where(ytop, y_low.to(y.dtype), y)
        x = torch.where(xtop, x_low.to(x.dtype), x)
        ly, lx = y - y_low, x - x_low
        hy, hx = 1.0 - ly, 1.0 - lx
        base = (bidx[sel] * H * W)[:, None, None]
        valid = (vy[:, :, None] & vx[:, None, :]).to(input.dtype)

        def tap(yy, xx, wy, wx):
            idx = base + yy[:, :, None] * W + xx[:, None, :]                      # (k, Py, Px)
            v = flat[idx.reshape(-1)].reshape(len(sel), P * g_h, P * g_w, C)
            return v * (wy[:, :, None] * wx[:, None, :] * valid)[..., None]

        val = tap(y_low, x_low, hy, hx) + tap(y_low, x_high, hy, lx) + tap(y_high, x_low, ly, hx) + tap(y_high, x_high, ly, lx)
        val = val.reshape(len(sel), P, g_h, P, g_w, C).sum(dim=(2, 4)) / count   # (k, P, P, C)
        out = out.index_put((sel,), val.permute(0, 3, 1, 2))
    return out


def roi_pool(input, rois, output_size, spatial_scale):
    """torchvision.ops.roi_pool restated (detectron2 POOLER_TYPE "ROIPool" builds RoIPool(output_size, spatial_scale) per level).
    input (N,C,H,W), rois (K,5) = [batch, x1, y1, x2, y2] -> (K,C,P,P), differentiable w.r.t. input.  Follows the published kernel:
    ROI corners rounded to whole pixels with C round() (half away from zero), a ROI of at least 1 x 1 (roi_end - roi_start + 1),
    bin (ph, pw) = rows [floor(ph * bh), ceil((ph + 1) * bh)) + roi_start clipped to [0, H] (bh = roi_height / P in float32), same for
    columns; strictly-greater scan in row-major order (the FIRST maximum), an empty bin is 0 and passes no gradient."""
    P = output_size if isinstance(output_size, int) else output_size[0]
    N, C, H, W = input.shape
    K = rois.shape[0]
    rows = []

    def c_round(v):          # C round(): half away from zero (torch.round is half to even)
        v = float(v)
        return int(math.floor(abs(v) + 0.5)) * (1 if v >= 0 else -1)
    scale32 = torch.tensor(spatial_scale, dtype=torch.float32)
    for k in range(K):
        b = int(rois[k, 0])
        rsw, rsh, rew, reh = (c_round(rois[k, i].float() * scale32) for i in (1, 2, 3, 4))       # (the kernel's arithmetic is float32)
        rw, rh = max(rew - rsw + 1, 1), max(reh - rsh + 1, 1)
        bh = torch.tensor(float(rh), dtype=torch.float32) / torch.tensor(float(P), dtype=torch.float32)
        bw = torch.tensor(float(rw), dtype=torch.float32) / torch.tensor(float(P), dtype=torch.float32)
        bins = []
        for ph in range(P):
            for pw in range(P):
                hs = int(torch.floor(torch.tensor(float(ph), dtype=torch.float32) * bh)) + rsh
                he = int(torch.ceil(torch.tensor(float(ph + 1), dtype=torch.float32) * bh)) + rsh
                ws = int(torch.floor(torch.tensor(float(pw), dtype=torch.float32) * bw)) + rsw
                we = int(torch.ceil(torch.tensor(float(pw + 1), dtype=torch.float32) * bw)) + rsw
                hs, he = min(max(hs, 0), H), min(max(he, 0), H)
                ws, we = min(max(ws, 0), W), min(max(we, 0), W)
                if he <= hs or we <= ws:
                    bins.append(input.new_zeros((C,)))
                    continue
                region = input[b, :, hs:he, ws:we].reshape(C, -1)
                idx = region.argmax(dim=1, keepdim=True)             # first occurrence of the maximum (row-major scan)
                bins.append(region.gather(1, idx)[:, 0])
        rows.append(torch.stack(bins, dim=1).reshape(C, P, P))
    return torch.stack(rows) if rows else input.new_zeros((0, C, P, P))


def assign_boxes_to_levels(box_lists, min_level, max_level, canonical_box_size, canonical_level):
    box_sizes = torch.sqrt(cat([boxes.area() for boxes in box_lists]))
    level_assignments = torch.floor(canonical_level + torch.log2(box_sizes / canonical_box_size + 1e-8))
    level_assignments = torch.clamp(level_assignments, min=min_level, max=max_level)
    return level_assignments.to(torch.int64) - min_level


def convert_boxes_to_pooler_format(box_lists):
    boxes = torch.cat([x.tensor for x in box_lists], dim=0)
    sizes = [len(b) for b in box_lists]
    indices = torch.repeat_interleave(torch.arange(len(box_lists), dtype=boxes.dtype, device=boxes.device),
                                      torch.tensor(sizes, device=boxes.device))
    return cat([indices[:, None], boxes], dim=1)


class ROIPooler(nn.Module):
    def __init__(self, output_size, scales, sampling_ratio, pooler_type, canonical_box_size=224, canonical_level=4):
        super().__init__()
        if isinstance(output_size, int):
            output_size = (output_size, output_size)
        assert pooler_type == "ROIAlignV2", pooler_type
        self.output_size = output_size
        self.scales = scales
        self.sampling_ratio = sampling_ratio
        min_level = -(math.log2(scales[0]))
        max_level = -(math.log2(scales[-1]))
        assert math.isclose(min_level, int(min_level)) and math.isclose(max_level, int(max_level))
        self.min_level, self.max_level = int(min_level), int(max_level)
        assert len(scales) == self.max_level - self.min_level + 1
        self.canonical_level = canonical_level
        self.canonical_box_size = canonical_box_size

    def forward(self, x: List[torch.Tensor], box_lists: List[Boxes]):
        pooler_fmt_boxes = convert_boxes_to_pooler_format(box_lists)
        if len(self.scales) == 1:
            return roi_align(x[0], pooler_fmt_boxes, self.output_size, self.scales[0], self.sampling_ratio, True)
        level_assignments = assign_boxes_to_levels(box_lists, self.min_level, self.max_level, self.canonical_box_size,
                                                   self.canonical_level)
        num_boxes = pooler_fmt_boxes.size(0)
        output = torch.zeros((num_boxes, x[0].shape[1], self.output_size[0], self.output_size[0]), dtype=x[0].dtype)
        for level, scale in enumerate(self.scales):
            inds = nonzero_tuple(level_assignments == level)[0]
            if inds.numel() == 0:
                continue
            output = output.index_put((inds,), roi_align(x[level], pooler_fmt_boxes[inds], self.output_size, scale,
                                                         self.sampling_ratio, True))
        return output


# ------------------------------------------------------------------------------------------------
# detectron2.modeling.backbone: Backbone, FPN, LastLevelMaxPool  (dla.py:13-15,500-506; resnet.py:88-95)
# ------------------------------------------------------------------------------------------------

BACKBONE_REGISTRY = Registry("BACKBONE")


class Backbone(nn.Module):
    @property
    def size_divisibility(self) -> int:
        return 0

    def output_shape(self):
        return {name: ShapeSpec(channels=self._out_feature_channels[name], stride=self._out_feature_strides[name])
                for name in self._out_features}


class LastLevelMaxPool(nn.Module):
    def __init__(self):
        super().__init__()
        self.num_levels = 1
        self.in_feature = "p5"

    def forward(self, x):
        return [F.max_pool2d(x, kernel_size=1, stride=2, padding=0)]


class FPN(Backbone):
    def __init__(self, bottom_up, in_features, out_channels, norm="", top_block=None, fuse_type="sum"):
        super().__init__()
        assert norm == "" and fuse_type in ("sum", "avg")
        input_shapes = bottom_up.output_shape()
        strides = [input_shapes[f].stride for f in in_features]
        in_channels_per_feature = [input_shapes[f].channels for f in in_features]
        lateral_convs, output_convs = [], []
        for idx, in_channels in enumerate(in_channels_per_feature):
            lateral_conv = nn.Conv2d(in_channels, out_channels, kernel_size=1, bias=True)
            output_conv = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=True)
            c2_xavier_fill(lateral_conv)
            c2_xavier_fill(output_conv)
            stage = int(math.log2(strides[idx]))
            self.add_module(f"fpn_lateral{stage}", lateral_conv)
            self.add_module(f"fpn_output{stage}", output_conv)
            lateral_convs.append(lateral_conv)
            output_convs.append(output_conv)
        self.lateral_convs = lateral_convs[::-1]
        self.output_convs = output_convs[::-1]
        self.top_block = top_block
        self.in_features = tuple(in_features)
        self.bottom_up = bottom_up
        self._out_feature_strides = {f"p{int(math.log2(s))}": s for s in strides}
        if self.top_block is not None:
            for s in range(stage, stage + self.top_block.num_levels):
                self._out_feature_strides[f"p{s + 1}"] = 2 ** (s + 1)
        self._out_features = list(self._out_feature_strides.keys())
        self._out_feature_channels = {k: out_channels for k in self._out_features}
        self._size_divisibility = strides[-1]
        self._fuse_type = fuse_type

    @property
    def size_divisibility(self):
        return self._size_divisibility

    def forward(self, x):
        bottom_up_features = self.bottom_up(x)
        results = []
        prev_features = self.lateral_convs[0](bottom_up_features[self.in_features[-1]])
        results.append(self.output_convs[0](prev_features))
        for idx, (lateral_conv, output_conv) in enumerate(zip(self.lateral_convs, self.output_convs)):
            if idx > 0:
                features = bottom_up_features[self.in_features[-idx - 1]]
                top_down_features = F.interpolate(prev_features, scale_factor=2.0, mode="nearest")
                prev_features = lateral_conv(features) + top_down_features
                if self._fuse_type == "avg":
                    prev_features = prev_features / 2
                results.insert(0, output_conv(prev_features))
        if self.top_block is not None:
            if self.top_block.in_feature in bottom_up_features:
                top_block_in_feature = bottom_up_features[self.top_block.in_feature]
            else:
                top_block_in_feature = results[self._out_features.index(self.top_block.in_feature)]
            results.extend(self.top_block(top_block_in_feature))
        assert len(self._out_features) == len(results)
        return {f: res for f, res in zip(self._out_features, results)}


# ------------------------------------------------------------------------------------------------
# torchvision.models.resnet18/34 (BasicBlock variants), the `base` of the reference's ResNet wrapper
#   (cubercnn/modeling/backbone/resnet.py:2,16-20,30-37).  torchvision is a pip dependency that is
#   absent from /root/reference and from this image; restated from its published architecture
#   (He et al. 2016; torchvision/models/resnet.py: conv7x7/s2 - BN - ReLU - maxpool3x3/s2/p1 -
#   [3,4,6,3] BasicBlocks, stride on conv1 of the first block of layer2..4 with a 1x1/s2 conv+BN
#   `downsample`, conv weights kaiming_normal_(fan_out, relu), BN weight 1 / bias 0).
# ------------------------------------------------------------------------------------------------


class TVBasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class TVBottleneck(nn.Module):
    """torchvision Bottleneck (v1.5: the stride sits on the 3x3), expansion 4"""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, 1, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class TVResNet(nn.Module):
    def __init__(self, layers, block=None):
        super().__init__()
        self.block = block or TVBasicBlock
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0], 1)
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * self.block.expansion, 1000)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride):
        block, downsample = self.block, None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                       nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)


def _tv_resnet(layers, pretrained, block=None):
    if pretrained:
        raise RuntimeError("ImageNet weights are a network download (torchvision); set MODEL.WEIGHTS")
    return TVResNet(layers, block)


def tv_resnet50(pretrained=False):
    """published parameter count 25 557 032"""
    return _tv_resnet([3, 4, 6, 3], pretrained, TVBottleneck)


def tv_resnet101(pretrained=False):
    """published parameter count 44 549 160"""
    return _tv_resnet([3, 4, 23, 3], pretrained, TVBottleneck)


def tv_resnet18(pretrained=False):
    return _tv_resnet([2, 2, 2, 2], pretrained)


def tv_resnet34(pretrained=False):
    return _tv_resnet([3, 4, 6, 3], pretrained)


# torchvision.models.densenet121 (cubercnn/modeling/backbone/densenet.py:2,14-15) -- un-vendored, restated from the published
# architecture (Huang et al.; torchvision/models/densenet.py): growth 32, blocks (6, 12, 24, 16), bn_size 4, 64 stem features,
# drop rate 0; `features` = conv0 7x7/s2, norm0, relu0, pool0 3x3/s2, denseblock1, transition1, ..., denseblock4, norm5.
# PARITY UNPINNED: no torchvision binary or fixture exists here to check this restatement against.
class TVDenseLayer(nn.Module):
    def __init__(self, cin, growth, bn_size):
        super().__init__()
        self.norm1 = nn.BatchNorm2d(cin)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(cin, bn_size * growth, kernel_size=1, stride=1, bias=False)
        self.norm2 = nn.BatchNorm2d(bn_size * growth)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(bn_size * growth, growth, kernel_size=3, stride=1, padding=1, bias=False)

    def forward(self, inputs):
        prev = [inputs] if isinstance(inputs, torch.Tensor) else inputs
        bottleneck = self.conv1(self.relu1(self.norm1(torch.cat(prev, 1))))
        return self.conv2(self.relu2(self.norm2(bottleneck)))


class TVDenseBlock(nn.ModuleDict):
    def __init__(self, num_layers, cin, bn_size, growth):
        super().__init__()
        for i in range(num_layers):
            self.add_module("denselayer%d" % (i + 1), TVDenseLayer(cin + i * growth, growth, bn_size))

    def forward(self, init_features):
        features = [init_features]
        for _, layer in self.items():
            features.append(layer(features))
        return torch.cat(features, 1)


class TVTransition(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__()
        self.add_module("norm", nn.BatchNorm2d(cin))
        self.add_module("relu", nn.ReLU(inplace=True))
        self.add_module("conv", nn.Conv2d(cin, cout, kernel_size=1, stride=1, bias=False))
        self.add_module("pool", nn.AvgPool2d(kernel_size=2, stride=2))


class TVDenseNet(nn.Module):
    def __init__(self, growth=32, block_config=(6, 12, 24, 16), init_features=64, bn_size=4, num_classes=1000):
        super().__init__()
        from collections import OrderedDict
        self.features = nn.Sequential(OrderedDict([
            ("conv0", nn.Conv2d(3, init_features, kernel_size=7, stride=2, padding=3, bias=False)),
            ("norm0", nn.BatchNorm2d(init_features)), ("relu0", nn.ReLU(inplace=True)),
            ("pool0", nn.MaxPool2d(kernel_size=3, stride=2, padding=1))]))
        c = init_features
        for i, n in enumerate(block_config):
            self.features.add_module("denseblock%d" % (i + 1), TVDenseBlock(n, c, bn_size, growth))
            c += n * growth
            if i != len(block_config) - 1:
                self.features.add_module("transition%d" % (i + 1), TVTransition(c, c // 2))
                c //= 2
        self.features.add_module("norm5", nn.BatchNorm2d(c))
        self.classifier = nn.Linear(c, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.constant_(m.bias, 0)


def tv_densenet121(pretrained=False):
    if pretrained:
        raise RuntimeError("ImageNet weights are a network download (torchvision); set MODEL.WEIGHTS")
    return TVDenseNet()


# torchvision.models.mnasnet1_0 (cubercnn/modeling/backbone/mnasnet.py:2,14-15) -- un-vendored, restated from the published
# architecture (Tan et al.; torchvision/models/mnasnet.py): depths [32, 16, 24, 40, 80, 96, 192, 320], inverted-residual stacks
# (k3 s2 x3 x3), (k5 s2 x3 x3), (k5 s2 x6 x3), (k3 s1 x6 x2), (k5 s2 x6 x4), (k3 s1 x6 x1), BN momentum 1 - 0.9997.
# Its parameter count reproduces torchvision's published 4 383 312; beyond that PARITY UNPINNED (no torchvision binary here).
_MNAS_BN_MOMENTUM = 1 - 0.9997


class TVInvertedResidual(nn.Module):
    def __init__(self, in_ch, out_ch, kernel_size, stride, expansion_factor, bn_momentum=0.1):
        super().__init__()
        mid_ch = in_ch * expansion_factor
        self.apply_residual = in_ch == out_ch and stride == 1
        self.layers = nn.Sequential(
            nn.Conv2d(in_ch, mid_ch, 1, bias=False), nn.BatchNorm2d(mid_ch, momentum=bn_momentum), nn.ReLU(inplace=True),
            nn.Conv2d(mid_ch, mid_ch, kernel_size, padding=kernel_size // 2, stride=stride, groups=mid_ch, bias=False),
            nn.BatchNorm2d(mid_ch, momentum=bn_momentum), nn.ReLU(inplace=True),
            nn.Conv2d(mid_ch, out_ch, 1, bias=False), nn.BatchNorm2d(out_ch, momentum=bn_momentum))

    def forward(self, input):
        return self.layers(input) + input if self.apply_residual else self.layers(input)


def _tv_mnas_stack(in_ch, out_ch, kernel_size, stride, exp_factor, repeats, bn_momentum):
    first = TVInvertedResidual(in_ch, out_ch, kernel_size, stride, exp_factor, bn_momentum=bn_momentum)
    rest = [TVInvertedResidual(out_ch, out_ch, kernel_size, 1, exp_factor, bn_momentum=bn_momentum) for _ in range(1, repeats)]
    return nn.Sequential(first, *rest)


class TVMNASNet(nn.Module):
    def __init__(self, num_classes=1000, dropout=0.2):
        super().__init__()
        d, m = [32, 16, 24, 40, 80, 96, 192, 320], _MNAS_BN_MOMENTUM
        self.layers = nn.Sequential(
            nn.Conv2d(3, d[0], 3, padding=1, stride=2, bias=False), nn.BatchNorm2d(d[0], momentum=m), nn.ReLU(inplace=True),
            nn.Conv2d(d[0], d[0], 3, padding=1, stride=1, groups=d[0], bias=False), nn.BatchNorm2d(d[0], momentum=m), nn.ReLU(inplace=True),
            nn.Conv2d(d[0], d[1], 1, padding=0, stride=1, bias=False), nn.BatchNorm2d(d[1], momentum=m),
            _tv_mnas_stack(d[1], d[2], 3, 2, 3, 3, m), _tv_mnas_stack(d[2], d[3], 5, 2, 3, 3, m), _tv_mnas_stack(d[3], d[4], 5, 2, 6, 3, m),
            _tv_mnas_stack(d[4], d[5], 3, 1, 6, 2, m), _tv_mnas_stack(d[5], d[6], 5, 2, 6, 4, m), _tv_mnas_stack(d[6], d[7], 3, 1, 6, 1, m),
            nn.Conv2d(d[7], 1280, 1, padding=0, stride=1, bias=False), nn.BatchNorm2d(1280, momentum=m), nn.ReLU(inplace=True))
        self.classifier = nn.Sequential(nn.Dropout(p=dropout, inplace=True), nn.Linear(1280, num_classes))
        for mod in self.modules():
            if isinstance(mod, nn.Conv2d):
                nn.init.kaiming_normal_(mod.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(mod, nn.BatchNorm2d):
                nn.init.ones_(mod.weight)
                nn.init.zeros_(mod.bias)
            elif isinstance(mod, nn.Linear):
                nn.init.kaiming_uniform_(mod.weight, mode="fan_out", nonlinearity="sigmoid")
                nn.init.zeros_(mod.bias)


def tv_mnasnet1_0(pretrained=False):
    if pretrained:
        raise RuntimeError("ImageNet weights are a network download (torchvision); set MODEL.WEIGHTS")
    return TVMNASNet()


# torchvision.models.shufflenet_v2_x1_0 (cubercnn/modeling/backbone/shufflenet.py:2,14-20) -- un-vendored, restated from the
# published architecture (Ma et al.; torchvision/models/shufflenetv2.py): repeats [4, 8, 4], channels [24, 116, 232, 464, 1024],
# PyTorch default initialisation.  Its parameter count reproduces torchvision's published 2 278 604; beyond that PARITY UNPINNED.
def tv_channel_shuffle(x, groups):
    batchsize, num_channels, height, width = x.size()
    x = x.view(batchsize, groups, num_channels // groups, height, width)
    x = torch.transpose(x, 1, 2).contiguous()
    return x.view(batchsize, -1, height, width)


class TVShuffleUnit(nn.Module):
    def __init__(self, inp, oup, stride):
        super().__init__()
        self.stride = stride
        bf = oup // 2
        dw = lambda c, s: nn.Conv2d(c, c, 3, s, 1, bias=False, groups=c)      # noqa: E731
        if stride > 1:
            self.branch1 = nn.Sequential(dw(inp, stride), nn.BatchNorm2d(inp), nn.Conv2d(inp, bf, 1, 1, 0, bias=False), nn.BatchNorm2d(bf),
                                         nn.ReLU(inplace=True))
        else:
            self.branch1 = nn.Sequential()
        self.branch2 = nn.Sequential(nn.Conv2d(inp if stride > 1 else bf, bf, 1, 1, 0, bias=False), nn.BatchNorm2d(bf), nn.ReLU(inplace=True),
                                     dw(bf, stride), nn.BatchNorm2d(bf), nn.Conv2d(bf, bf, 1, 1, 0, bias=False), nn.BatchNorm2d(bf),
                                     nn.ReLU(inplace=True))

    def forward(self, x):
        if self.stride == 1:
            x1, x2 = x.chunk(2, dim=1)
            out = torch.cat((x1, self.branch2(x2)), dim=1)
        else:
            out = torch.cat((self.branch1(x), self.branch2(x)), dim=1)
        return tv_channel_shuffle(out, 2)


class TVShuffleNetV2(nn.Module):
    def __init__(self, stages_repeats=(4, 8, 4), stages_out_channels=(24, 116, 232, 464, 1024), num_classes=1000):
        super().__init__()
        c = stages_out_channels[0]
        self.conv1 = nn.Sequential(nn.Conv2d(3, c, 3, 2, 1, bias=False), nn.BatchNorm2d(c), nn.ReLU(inplace=True))
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        for name, repeats, out in zip(("stage2", "stage3", "stage4"), stages_repeats, stages_out_channels[1:]):
            setattr(self, name, nn.Sequential(TVShuffleUnit(c, out, 2), *[TVShuffleUnit(out, out, 1) for _ in range(repeats - 1)]))
            c = out
        out = stages_out_channels[-1]
        self.conv5 = nn.Sequential(nn.Conv2d(c, out, 1, 1, 0, bias=False), nn.BatchNorm2d(out), nn.ReLU(inplace=True))
        self.fc = nn.Linear(out, num_classes)


def tv_shufflenet_v2_x1_0(pretrained=False):
    if pretrained:
        raise RuntimeError("ImageNet weights are a network download (torchvision); set MODEL.WEIGHTS")
    return TVShuffleNetV2()


def build_resnet_backbone(cfg, input_shape):
    raise NotImplementedError("MSRA ResNet (MODEL.RESNETS.TORCHVISION False) is outside the restated surface")


# ------------------------------------------------------------------------------------------------
# detectron2.modeling.roi_heads: ROIHeads / StandardROIHeads / box head / FastRCNNOutputLayers
#   (base classes of ROIHeads3D roi_heads.py:17-19,40 and FastRCNNOutputs fast_rcnn.py:11-13,119)
# ------------------------------------------------------------------------------------------------

ROI_HEADS_REGISTRY = Registry("ROI_HEADS")
ROI_BOX_HEAD_REGISTRY = Registry("ROI_BOX_HEAD")


def select_foreground_proposals(proposals, bg_label):
    assert isinstance(proposals, (list, tuple)) and isinstance(proposals[0], Instances)
    assert proposals[0].has("gt_classes")
    fg_proposals, fg_selection_masks = [], []
    for proposals_per_image in proposals:
        gt_classes = proposals_per_image.gt_classes
        fg_selection_mask = (gt_classes != -1) & (gt_classes != bg_label)
        fg_idxs = fg_selection_mask.nonzero().squeeze(1)
        fg_proposals.append(proposals_per_image[fg_idxs])
        fg_selection_masks.append(fg_selection_mask)
    return fg_proposals, fg_selection_masks


@ROI_BOX_HEAD_REGISTRY.register()
class FastRCNNConvFCHead(nn.Sequential):
    @configurable
    def __init__(self, input_shape: ShapeSpec, *, conv_dims, fc_dims, conv_norm=""):
        super().__init__()
        assert len(conv_dims) + len(fc_dims) > 0 and len(conv_dims) == 0
        self._output_size = (input_shape.channels, input_shape.height, input_shape.width)
        self.fcs = []
        for k, fc_dim in enumerate(fc_dims):
            if k == 0:
                self.add_module("flatten", nn.Flatten())
            fc = nn.Linear(int(np.prod(self._output_size)), fc_dim)
            self.add_module(f"fc{k + 1}", fc)
            self.add_module(f"fc_relu{k + 1}", nn.ReLU())
            self.fcs.append(fc)
            self._output_size = fc_dim
        for layer in self.fcs:
            c2_xavier_fill(layer)

    @classmethod
    def from_config(cls, cfg, input_shape):
        num_conv = cfg.MODEL.ROI_BOX_HEAD.NUM_CONV
        conv_dim = cfg.MODEL.ROI_BOX_HEAD.CONV_DIM
        num_fc = cfg.MODEL.ROI_BOX_HEAD.NUM_FC
        fc_dim = cfg.MODEL.ROI_BOX_HEAD.FC_DIM
        return {"input_shape": input_shape, "conv_dims": [conv_dim] * num_conv, "fc_dims": [fc_dim] * num_fc,
                "conv_norm": cfg.MODEL.ROI_BOX_HEAD.NORM}

    def forward(self, x):
        for layer in self:
            x = layer(x)
        return x

    @property
    def output_shape(self):
        o = self._output_size
        return ShapeSpec(channels=o) if isinstance(o, int) else ShapeSpec(channels=o[0], height=o[1], width=o[2])


def build_box_head(cfg, input_shape):
    return ROI_BOX_HEAD_REGISTRY.get(cfg.MODEL.ROI_BOX_HEAD.NAME)(cfg, input_shape)


def _log_classification_stats(pred_logits, gt_classes, prefix="fast_rcnn"):
    num_instances = gt_classes.numel()
    if num_instances == 0:
        return
    pred_classes = pred_logits.argmax(dim=1)
    bg_class_ind = pred_logits.shape[1] - 1
    fg_inds = (gt_classes >= 0) & (gt_classes < bg_class_ind)
    num_fg = fg_inds.nonzero().numel()
    fg_gt_classes = gt_classes[fg_inds]
    fg_pred_classes = pred_classes[fg_inds]
    num_false_negative = (fg_pred_classes == bg_class_ind).nonzero().numel()
    num_accurate = (pred_classes == gt_classes).nonzero().numel()
    fg_num_accurate = (fg_pred_classes == fg_gt_classes).nonzero().numel()
    storage = get_event_storage()
    storage.put_scalar(f"{prefix}/cls_accuracy", num_accurate / num_instances)
    if num_fg > 0:
        storage.put_scalar(f"{prefix}/fg_cls_accuracy", fg_num_accurate / num_fg)
        storage.put_scalar(f"{prefix}/false_negative", num_false_negative / num_fg)


class FastRCNNOutputLayers(nn.Module):
    @configurable
    def __init__(self, input_shape: ShapeSpec, *, box2box_transform, num_classes, test_score_thresh=0.0,
                 test_nms_thresh=0.5, test_topk_per_image=100, cls_agnostic_bbox_reg=False, smooth_l1_beta=0.0,
                 box_reg_loss_type="smooth_l1", loss_weight=1.0):
        super().__init__()
        if isinstance(input_shape, int):
            input_shape = ShapeSpec(channels=input_shape)
        self.num_classes = num_classes
        input_size = input_shape.channels * (input_shape.width or 1) * (input_shape.height or 1)
        self.cls_score = nn.Linear(input_size, num_classes + 1)
        num_bbox_reg_classes = 1 if cls_agnostic_bbox_reg else num_classes
        box_dim = len(box2box_transform.weights)
        self.bbox_pred = nn.Linear(input_size, num_bbox_reg_classes * box_dim)
        nn.init.normal_(self.cls_score.weight, std=0.01)
        nn.init.normal_(self.bbox_pred.weight, std=0.001)
        for l in [self.cls_score, self.bbox_pred]:
            nn.init.constant_(l.bias, 0)
        self.box2box_transform = box2box_transform
        self.smooth_l1_beta = smooth_l1_beta
        self.test_score_thresh = test_score_thresh
        self.test_nms_thresh = test_nms_thresh
        self.test_topk_per_image = test_topk_per_image
        self.box_reg_loss_type = box_reg_loss_type
        if isinstance(loss_weight, float):
            loss_weight = {"loss_cls": loss_weight, "loss_box_reg": loss_weight}
        self.loss_weight = loss_weight

    @classmethod
    def from_config(cls, cfg, input_shape):
        return {
            "input_shape": input_shape,
            "box2box_transform": Box2BoxTransform(weights=cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS),
            "num_classes": cfg.MODEL.ROI_HEADS.NUM_CLASSES,
            "cls_agnostic_bbox_reg": cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG,
            "smooth_l1_beta": cfg.MODEL.ROI_BOX_HEAD.SMOOTH_L1_BETA,
            "test_score_thresh": cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST,
            "test_nms_thresh": cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST,
            "test_topk_per_image": cfg.TEST.DETECTIONS_PER_IMAGE,
            "box_reg_loss_type": cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_TYPE,
            "loss_weight": {"loss_box_reg": cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_WEIGHT},
        }

    def forward(self, x):
        if x.dim() > 2:
            x = torch.flatten(x, start_dim=1)
        return self.cls_score(x), self.bbox_pred(x)

    def predict_boxes_for_gt_classes(self, predictions, proposals):
        if not len(proposals):
            return []
        scores, proposal_deltas = predictions
        proposal_boxes = cat([p.proposal_boxes.tensor for p in proposals], dim=0)
        N, B = proposal_boxes.shape
        predict_boxes = self.box2box_transform.apply_deltas(proposal_deltas, proposal_boxes)
        K = predict_boxes.shape[1] // B
        if K > 1:
            gt_classes = torch.cat([p.gt_classes for p in proposals], dim=0)
            gt_classes = gt_classes.clamp_(0, K - 1)
            predict_boxes = predict_boxes.view(N, K, B)[torch.arange(N, dtype=torch.long, device=predict_boxes.device), gt_classes]
        return predict_boxes.split([len(p) for p in proposals])

    def predict_boxes(self, predictions, proposals):
        if not len(proposals):
            return []
        _, proposal_deltas = predictions
        proposal_boxes = cat([p.proposal_boxes.tensor for p in proposals], dim=0)
        predict_boxes = self.box2box_transform.apply_deltas(proposal_deltas, proposal_boxes)
        return predict_boxes.split([len(p) for p in proposals])

    def predict_probs(self, predictions, proposals):
        scores, _ = predictions
        return F.softmax(scores, dim=-1).split([len(p) for p in proposals], dim=0)


class ROIHeads(nn.Module):
    @configurable
    def __init__(self, *, num_classes, batch_size_per_image, positive_fraction, proposal_matcher, proposal_append_gt=True):
        super().__init__()
        self.batch_size_per_image = batch_size_per_image
        self.positive_fraction = positive_fraction
        self.num_classes = num_classes
        self.proposal_matcher = proposal_matcher
        self.proposal_append_gt = proposal_append_gt

    @classmethod
    def from_config(cls, cfg):
        return {
            "batch_size_per_image": cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE,
            "positive_fraction": cfg.MODEL.ROI_HEADS.POSITIVE_FRACTION,
            "num_classes": cfg.MODEL.ROI_HEADS.NUM_CLASSES,
            "proposal_append_gt": cfg.MODEL.ROI_HEADS.PROPOSAL_APPEND_GT,
            "proposal_matcher": Matcher(cfg.MODEL.ROI_HEADS.IOU_THRESHOLDS, cfg.MODEL.ROI_HEADS.IOU_LABELS,
                                        allow_low_quality_matches=False),
        }


@ROI_HEADS_REGISTRY.register()
class StandardROIHeads(ROIHeads):
    @configurable
    def __init__(self, *, box_in_features, box_pooler, box_head, box_predictor, mask_in_features=None, mask_pooler=None,
                 mask_head=None, keypoint_in_features=None, keypoint_pooler=None, keypoint_head=None,
                 train_on_pred_boxes=False, **kwargs):
        super().__init__(**kwargs)
        self.in_features = self.box_in_features = box_in_features
        self.box_pooler = box_pooler
        self.box_head = box_head
        self.box_predictor = box_predictor
        self.mask_on = mask_in_features is not None
        self.keypoint_on = keypoint_in_features is not None
        self.train_on_pred_boxes = train_on_pred_boxes

    @classmethod
    def from_config(cls, cfg, input_shape):
        ret = super().from_config(cfg)
        ret["train_on_pred_boxes"] = cfg.MODEL.ROI_BOX_HEAD.TRAIN_ON_PRED_BOXES
        if inspect.ismethod(cls._init_box_head):
            ret.update(cls._init_box_head(cfg, input_shape))
        return ret

    @classmethod
    def _init_box_head(cls, cfg, input_shape):
        in_features = cfg.MODEL.ROI_HEADS.IN_FEATURES
        pooler_resolution = cfg.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION
        pooler_scales = tuple(1.0 / input_shape[k].stride for k in in_features)
        sampling_ratio = cfg.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO
        pooler_type = cfg.MODEL.ROI_BOX_HEAD.POOLER_TYPE
        in_channels = [input_shape[f].channels for f in in_features]
        assert len(set(in_channels)) == 1, in_channels
        in_channels = in_channels[0]
        box_pooler = ROIPooler(output_size=pooler_resolution, scales=pooler_scales, sampling_ratio=sampling_ratio,
                               pooler_type=pooler_type)
        box_head = build_box_head(cfg, ShapeSpec(channels=in_channels, height=pooler_resolution, width=pooler_resolution))
        box_predictor = FastRCNNOutputLayers(cfg, box_head.output_shape)
        return {"box_in_features": in_features, "box_pooler": box_pooler, "box_head": box_head, "box_predictor": box_predictor}


# ------------------------------------------------------------------------------------------------
# detectron2.modeling.meta_arch.GeneralizedRCNN  (base of RCNN3D, rcnn3d.py:15-17,26,46,110)
# ------------------------------------------------------------------------------------------------

META_ARCH_REGISTRY = Registry("META_ARCH")


def detector_postprocess(results: Instances, output_height: int, output_width: int):
    new_size = (output_height, output_width)
    scale_x, scale_y = output_width / results.image_size[1], output_height / results.image_size[0]
    results = Instances(new_size, **results.get_fields())
    output_boxes = results.pred_boxes if results.has("pred_boxes") else results.proposal_boxes
    output_boxes.scale(scale_x, scale_y)
    output_boxes.clip(results.image_size)
    return results[output_boxes.nonempty()]


class GeneralizedRCNN(nn.Module):
    @configurable
    def __init__(self, *, backbone, proposal_generator, roi_heads, pixel_mean, pixel_std, input_format=None, vis_period=0):
        super().__init__()
        self.backbone = backbone
        self.proposal_generator = proposal_generator
        self.roi_heads = roi_heads
        self.input_format = input_format
        self.vis_period = vis_period
        self.register_buffer("pixel_mean", torch.tensor(pixel_mean).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(pixel_std).view(-1, 1, 1), False)

    @property
    def device(self):
        return self.pixel_mean.device

    def preprocess_image(self, batched_inputs):
        images = [x["image"].to(self.device) for x in batched_inputs]
        images = [(x - self.pixel_mean) / self.pixel_std for x in images]
        return ImageList.from_tensors(images, self.backbone.size_divisibility)

    @staticmethod
    def _postprocess(instances, batched_inputs, image_sizes):
        processed_results = []
        for results_per_image, input_per_image, image_size in zip(instances, batched_inputs, image_sizes):
            height = input_per_image.get("height", image_size[0])
            width = input_per_image.get("width", image_size[1])
            processed_results.append({"instances": detector_postprocess(results_per_image, height, width)})
        return processed_results


# ------------------------------------------------------------------------------------------------
# pytorch3d.transforms  (cube_head.py:10-15,176; math_util.py:34,620,676)
# ------------------------------------------------------------------------------------------------


def rotation_6d_to_matrix(d6: torch.Tensor) -> torch.Tensor:
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = F.normalize(a1, dim=-1)
    b2 = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
    b2 = F.normalize(b2, dim=-1)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-2)


def quaternion_to_matrix(quaternions: torch.Tensor) -> torch.Tensor:
    r, i, j, k = torch.unbind(quaternions, -1)
    two_s = 2.0 / (quaternions * quaternions).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(quaternions.shape[:-1] + (3, 3))


def axis_angle_to_quaternion(axis_angle: torch.Tensor) -> torch.Tensor:
    angles = torch.norm(axis_angle, p=2, dim=-1, keepdim=True)
    half_angles = angles * 0.5
    eps = 1e-6
    small_angles = angles.abs() < eps
    sin_half_angles_over_angles = torch.empty_like(angles)
    sin_half_angles_over_angles[~small_angles] = torch.sin(half_angles[~small_angles]) / angles[~small_angles]
    sin_half_angles_over_angles[small_angles] = 0.5 - (angles[small_angles] * angles[small_angles]) / 48
    return torch.cat([torch.cos(half_angles), axis_angle * sin_half_angles_over_angles], dim=-1)


def axis_angle_to_matrix(axis_angle: torch.Tensor) -> torch.Tensor:
    return quaternion_to_matrix(axis_angle_to_quaternion(axis_angle))


def _axis_angle_rotation(axis: str, angle: torch.Tensor) -> torch.Tensor:
    """pytorch3d.transforms.rotation_conversions._axis_angle_rotation (published algorithm)"""
    cos, sin = torch.cos(angle), torch.sin(angle)
    one, zero = torch.ones_like(angle), torch.zeros_like(angle)
    if axis == "X":
        flat = (one, zero, zero, zero, cos, -sin, zero, sin, cos)
    elif axis == "Y":
        flat = (cos, zero, sin, zero, one, zero, -sin, zero, cos)
    elif axis == "Z":
        flat = (cos, -sin, zero, sin, cos, zero, zero, zero, one)
    else:
        raise ValueError("letter must be either X, Y or Z.")
    return torch.stack(flat, -1).reshape(angle.shape + (3, 3))


def euler_angles_to_matrix(euler_angles: torch.Tensor, convention: str) -> torch.Tensor:
    """pytorch3d.transforms.euler_angles_to_matrix: product of the three axis rotations in convention order (cube_head.py:184-185)"""
    if euler_angles.dim() == 0 or euler_angles.shape[-1] != 3:
        raise ValueError("Invalid input euler angles.")
    if len(convention) != 3:
        raise ValueError("Convention must have 3 letters.")
    mats = [_axis_angle_rotation(c, e) for c, e in zip(convention, torch.unbind(euler_angles, -1))]
    return torch.matmul(torch.matmul(mats[0], mats[1]), mats[2])


def so3_relative_angle(R1, R2, cos_angle: bool = False, cos_bound: float = 1e-4, eps: float = 1e-4):
    """pytorch3d.transforms.so3.so3_relative_angle / so3_rotation_angle (published algorithm): the angle of R1 R2^T from its
    trace; raises when a trace leaves [-1 - eps, 3 + eps].  The reference only uses cos_angle=True (roi_heads.py:631-633)."""
    R12 = torch.bmm(R1, R2.permute(0, 2, 1))
    rot_trace = R12[:, 0, 0] + R12[:, 1, 1] + R12[:, 2, 2]
    if ((rot_trace < -1.0 - eps) + (rot_trace > 3.0 + eps)).any():
        raise ValueError("A matrix has trace outside valid range [-1-eps,3+eps].")
    phi_cos = (rot_trace - 1.0) * 0.5
    if cos_angle:
        return phi_cos
    return torch.acos(phi_cos.clamp(-1.0 + cos_bound, 1.0 - cos_bound))


def _copysign(a, b):
    signs_differ = (a < 0) != (b < 0)
    return torch.where(signs_differ, -a, a)


# pytorch3d.ops.iou_box3d index tables (omni3d_evaluation.py:40,70,94)
_box_planes = [[0, 1, 2, 3], [3, 2, 6, 7], [0, 1, 5, 4], [0, 3, 7, 4], [1, 2, 6, 5], [4, 5, 6, 7]]
_box_triangles = [[0, 1, 2], [0, 3, 2], [4, 5, 6], [4, 6, 7], [1, 5, 6], [1, 6, 2],
                  [0, 4, 7], [0, 7, 3], [3, 2, 6], [3, 6, 7], [0, 1, 5], [0, 4, 5]]


class _C:
    """pytorch3d._C stand-in backed by the C oracle (oracle/iou_box3d_oracle.c)."""

    @staticmethod
    def iou_box3d(boxes1, boxes2):
        import ctypes
        import os
        lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "liboracle.so"))
        b1 = np.ascontiguousarray(boxes1.detach().cpu().numpy(), np.float32)
        b2 = np.ascontiguousarray(boxes2.detach().cpu().numpy(), np.float32)
        N, M = len(b1), len(b2)
        vol = np.zeros((N, M), np.float32)
        iou = np.zeros((N, M), np.float32)
        P = ctypes.c_void_p
        lib.iou_box3d_oracle(b1.ctypes.data_as(P), N, b2.ctypes.data_as(P), M, vol.ctypes.data_as(P), iou.ctypes.data_as(P))
        return torch.from_numpy(vol), torch.from_numpy(iou)


# ------------------------------------------------------------------------------------------------
# detectron2.solver: WarmupMultiStepLR (tools/train_net.py:125)
# ------------------------------------------------------------------------------------------------


def warmup_multistep_lr_factor(it, steps, gamma=0.1, warmup_iters=1000, warmup_factor=0.001, method="linear"):
    from bisect import bisect_right
    f = gamma ** bisect_right(list(steps), it)
    if it < warmup_iters:
        if method == "constant":
            f *= warmup_factor
        else:
            alpha = it / warmup_iters
            f *= warmup_factor * (1 - alpha) + alpha
    return f


# ------------------------------------------------------------------------------------------------
# pycocotools.coco.COCO / pycocotools.mask.iou (pycocotools 2.0, not installable here): the calls
# cubercnn/data/datasets.py:140-448 and cubercnn/evaluation/omni3d_evaluation.py make, restated from
# the published API so that the reference's OWN dataset / evaluator files can be run as the oracle
# for omni3d_amd/cubercnn/data/datasets.py (tests/test_datasets_io.py).  TEST INFRASTRUCTURE.
# ------------------------------------------------------------------------------------------------
class COCO:
    def __init__(self, annotation_file=None):
        import json as _json
        from collections import defaultdict as _dd
        self.dataset, self.anns, self.cats, self.imgs = dict(), dict(), dict(), dict()
        self.imgToAnns, self.catToImgs = _dd(list), _dd(list)
        if annotation_file is not None:
            with open(annotation_file, "r") as f:
                dataset = _json.load(f)
            assert type(dataset) == dict, "annotation file format {} not supported".format(type(dataset))
            self.dataset = dataset
            self.createIndex()

    def createIndex(self):
        from collections import defaultdict as _dd
        anns, cats, imgs = {}, {}, {}
        imgToAnns, catToImgs = _dd(list), _dd(list)
        if "annotations" in self.dataset:
            for ann in self.dataset["annotations"]:
                imgToAnns[ann["image_id"]].append(ann)
                anns[ann["id"]] = ann
        if "images" in self.dataset:
            for img in self.dataset["images"]:
                imgs[img["id"]] = img
        if "categories" in self.dataset:
            for cat in self.dataset["categories"]:
                cats[cat["id"]] = cat
        if "annotations" in self.dataset and "categories" in self.dataset:
            for ann in self.dataset["annotations"]:
                catToImgs[ann["category_id"]].append(ann["image_id"])
        self.anns, self.imgToAnns, self.catToImgs, self.imgs, self.cats = anns, imgToAnns, catToImgs, imgs, cats

    @staticmethod
    def _isArrayLike(obj):
        return hasattr(obj, "__iter__") and hasattr(obj, "__len__")

    def getAnnIds(self, imgIds=[], catIds=[], areaRng=[], iscrowd=None):
        import itertools as _it
        imgIds = imgIds if self._isArrayLike(imgIds) else [imgIds]
        catIds = catIds if self._isArrayLike(catIds) else [catIds]
        if len(imgIds) == len(catIds) == len(areaRng) == 0:
            anns = self.dataset["annotations"]
        else:
            if not len(imgIds) == 0:
                lists = [self.imgToAnns[imgId] for imgId in imgIds if imgId in self.imgToAnns]
                anns = list(_it.chain.from_iterable(lists))
            else:
                anns = self.dataset["annotations"]
            anns = anns if len(catIds) == 0 else [ann for ann in anns if ann["category_id"] in catIds]
            anns = anns if len(areaRng) == 0 else [ann for ann in anns if ann["area"] > areaRng[0] and ann["area"] < areaRng[1]]
        if iscrowd is not None:
            return [ann["id"] for ann in anns if ann["iscrowd"] == iscrowd]
        return [ann["id"] for ann in anns]

    def getCatIds(self, catNms=[], supNms=[], catIds=[]):
        catNms = catNms if self._isArrayLike(catNms) else [catNms]
        supNms = supNms if self._isArrayLike(supNms) else [supNms]
        catIds = catIds if self._isArrayLike(catIds) else [catIds]
        if len(catNms) == len(supNms) == len(catIds) == 0:
            cats = self.dataset["categories"]
        else:
            cats = self.dataset["categories"]
            cats = cats if len(catNms) == 0 else [cat for cat in cats if cat["name"] in catNms]
            cats = cats if len(supNms) == 0 else [cat for cat in cats if cat["supercategory"] in supNms]
            cats = cats if len(catIds) == 0 else [cat for cat in cats if cat["id"] in catIds]
        return [cat["id"] for cat in cats]

    def getImgIds(self, imgIds=[], catIds=[]):
        imgIds = imgIds if self._isArrayLike(imgIds) else [imgIds]
        catIds = catIds if self._isArrayLike(catIds) else [catIds]
        if len(imgIds) == len(catIds) == 0:
            ids = self.imgs.keys()
        else:
            ids = set(imgIds)
            for i, catId in enumerate(catIds):
                if i == 0 and len(ids) == 0:
                    ids = set(self.catToImgs[catId])
                else:
                    ids &= set(self.catToImgs[catId])
        return list(ids)

    def loadAnns(self, ids=[]):
        return [self.anns[i] for i in ids] if self._isArrayLike(ids) else [self.anns[ids]]

    def loadCats(self, ids=[]):
        return [self.cats[i] for i in ids] if self._isArrayLike(ids) else [self.cats[ids]]

    def loadImgs(self, ids=[]):
        return [self.imgs[i] for i in ids] if self._isArrayLike(ids) else [self.imgs[ids]]

    def loadRes(self, resFile):
        import copy as _copy
        res = COCO()
        res.dataset["images"] = [img for img in self.dataset["images"]]
        anns = resFile
        assert type(anns) == list, "results in not an array of objects"
        annsImgIds = [ann["image_id"] for ann in anns]
        assert set(annsImgIds) == (set(annsImgIds) & set(self.getImgIds())), "Results do not correspond to current coco set"
        if "bbox" in anns[0] and not anns[0]["bbox"] == []:
            res.dataset["categories"] = _copy.deepcopy(self.dataset["categories"])
            for id, ann in enumerate(anns):
                bb = ann["bbox"]
                ann["area"] = bb[2] * bb[3]
                ann["id"] = id + 1
                ann["iscrowd"] = 0
        res.dataset["annotations"] = anns
        res.createIndex()
        return res


def coco_box_iou(dt, gt, iscrowd):
    """pycocotools.mask.iou for XYWH boxes: (D, G) array, [] when either list is empty (bbIou of maskApi.c)"""
    import numpy as _np
    if len(dt) == 0 or len(gt) == 0:
        return []
    d, g = _np.asarray(dt, dtype=_np.float64).reshape(-1, 4), _np.asarray(gt, dtype=_np.float64).reshape(-1, 4)
    out = _np.zeros((len(d), len(g)))
    for j in range(len(g)):
        ga = g[j, 2] * g[j, 3]
        for i in range(len(d)):
            da = d[i, 2] * d[i, 3]
            w = min(d[i, 0] + d[i, 2], g[j, 0] + g[j, 2]) - max(d[i, 0], g[j, 0])
            h = min(d[i, 1] + d[i, 3], g[j, 1] + g[j, 3]) - max(d[i, 1], g[j, 1])
            inter = max(w, 0) * max(h, 0)
            u = da if iscrowd[j] else da + ga - inter
            out[i, j] = inter / u if u > 0 else 0.0
    return out


class PathManagerLocal:
    """detectron2.utils.file_io.PathManager for local paths"""

    @staticmethod
    def get_local_path(path):
        return path

    @staticmethod
    def open(path, mode="r"):
        return open(path, mode)

    @staticmethod
    def mkdirs(path):
        import os as _os
        _os.makedirs(path, exist_ok=True)

    @staticmethod
    def exists(path):
        import os as _os
        return _os.path.exists(path)

    @staticmethod
    def register_handler(handler):        # cubercnn/util/model_zoo.py:25 registers its `cubercnn://` handler at import time
        pass


class Timer:
    """fvcore.common.timer.Timer (seconds() only)"""

    def __init__(self):
        import time as _time
        self._t0 = _time.perf_counter()

    def seconds(self):
        import time as _time
        return _time.perf_counter() - self._t0
