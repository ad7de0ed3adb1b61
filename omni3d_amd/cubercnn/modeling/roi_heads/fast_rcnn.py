"""`FastRCNNOutputs` (= detectron2 FastRCNNOutputLayers + the reference's losses / inference),
/root/reference/cubercnn/modeling/roi_heads/fast_rcnn.py:57-260.  `cls_score` (K+1) and `bbox_pred`
(4K) are evaluated as one fused GEMM; losses run in csrc/box_loss.hip."""
import torch
from torch import nn

from .... import functional as HF
from ....d2.config import configurable
from ....d2.layers import ShapeSpec
from ..layers import Linear


class FastRCNNOutputs(nn.Module):
    @configurable
    def __init__(self, input_shape, *, box2box_weights, num_classes, test_score_thresh=0.0, test_nms_thresh=0.5,
                 test_topk_per_image=100, cls_agnostic_bbox_reg=False, smooth_l1_beta=0.0, box_reg_loss_type="smooth_l1",
                 loss_weight=1.0):
        super().__init__()
        if isinstance(input_shape, int):
            input_shape = ShapeSpec(channels=input_shape)
        if cls_agnostic_bbox_reg or box_reg_loss_type != "smooth_l1" or smooth_l1_beta != 0.0:
            raise NotImplementedError("MI355X hot path: class-specific L1 box regression (reference defaults)")
        self.num_classes = num_classes
        input_size = input_shape.channels * (input_shape.width or 1) * (input_shape.height or 1)
        self.cls_score = Linear(input_size, num_classes + 1)
        self.bbox_pred = Linear(input_size, num_classes * 4)
        nn.init.normal_(self.cls_score.weight, std=0.01)
        nn.init.normal_(self.bbox_pred.weight, std=0.001)
        for l in (self.cls_score, self.bbox_pred):
            nn.init.constant_(l.bias, 0)
        self.box2box_weights = tuple(box2box_weights)
        self.test_score_thresh = test_score_thresh
        self.test_nms_thresh = test_nms_thresh
        self.test_topk_per_image = test_topk_per_image
        self.loss_weight = loss_weight if isinstance(loss_weight, dict) else {"loss_cls": loss_weight, "loss_box_reg": loss_weight}
        self.fused_dim = (5 * num_classes + 1 + 15) // 16 * 16
        self.pending_logs = {}

    @classmethod
    def from_config(cls, cfg, input_shape):
        return {
            "input_shape": input_shape, "box2box_weights": cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS,
            "num_classes": cfg.MODEL.ROI_HEADS.NUM_CLASSES, "cls_agnostic_bbox_reg": cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG,
            "smooth_l1_beta": cfg.MODEL.ROI_BOX_HEAD.SMOOTH_L1_BETA, "test_score_thresh": cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST,
            "test_nms_thresh": cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST, "test_topk_per_image": cfg.TEST.DETECTIONS_PER_IMAGE,
            "box_reg_loss_type": cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_TYPE,
            "loss_weight": {"loss_box_reg": cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_WEIGHT},
        }

    def forward(self, x):
        """x (R, 1024) -> fused predictions (R, fused_dim) = [K+1 logits | 4K deltas | pad]."""
        return HF.fused_linear(x, [self.cls_score.weight, self.bbox_pred.weight], [self.cls_score.bias, self.bbox_pred.bias], self.fused_dim)

    def fused_param_groups(self):
        """read by solver/build.py tag_fused_groups: the optimizer keeps these back to back (+ zero padding) in its buckets"""
        pad = self.fused_dim - (5 * self.num_classes + 1)
        return [([self.cls_score.weight, self.bbox_pred.weight], pad * self.cls_score.weight.shape[1]),
                ([self.cls_score.bias, self.bbox_pred.bias], pad)]

    def losses(self, pred, cls, prop_boxes, targets, gt_row):
        """fast_rcnn.py:145-194; cls (R) int32 with -2 on padding rows."""
        names = ("BoxHead/loss_cls", "BoxHead/loss_box_reg")
        w = tuple(self.loss_weight.get(k, 1.0) for k in names)
        vec, sums = HF.box_loss(pred, self.num_classes, cls, prop_boxes, targets.gt, gt_row, self.box2box_weights, w)
        self.pending_logs = {"fast_rcnn": sums}
        return HF.LossDict({names[0]: vec[0], names[1]: vec[1]}, vectors=[(vec, names)])

    def flush_logs(self, storage):
        if "fast_rcnn" in self.pending_logs:
            s = self.pending_logs.pop("fast_rcnn").tolist()
            n, nfg = max(s[2], 1.0), s[3]
            storage.put_scalar("fast_rcnn/cls_accuracy", s[4] / n)
            if nfg > 0:
                storage.put_scalar("fast_rcnn/fg_cls_accuracy", s[5] / nfg)
                storage.put_scalar("fast_rcnn/false_negative", s[6] / nfg)
