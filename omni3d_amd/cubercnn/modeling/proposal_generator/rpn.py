"""`RPNWithIgnore` (PROPOSAL_GENERATOR_REGISTRY) and detectron2's `StandardRPNHead`, on the HIP kernels.

Mirrors /root/reference/cubercnn/modeling/proposal_generator/rpn.py (RPNWithIgnore :19-204: anchor
labelling with ignore regions and IoU-weighted sampling :41-127, "IoUness" objectness + IoU-weighted
L1 box losses :129-273) and the detectron2 RPN it extends (head, anchors, decode, per-level top-k,
NMS, top-N: SURVEY.md A.3-A.7).  Same constructor / from_config keys, same loss names and logged
scalar names; the per-image Python loops, boolean-mask gathers and .item()/tolist() syncs of the
reference are replaced by batched kernels that keep everything on the device."""
import math

import torch
from torch import nn

from .... import functional as HF
from ....d2.config import configurable
from ....d2.structures import Boxes, Instances
from ....kernels import det, select
from ..anchors import build_anchor_generator
from ..layers import Conv2d
from ..registries import PROPOSAL_GENERATOR_REGISTRY, RPN_HEAD_REGISTRY

CL = torch.channels_last
import os as _os
_FUSED_HEAD = _os.environ.get("OMNI_RPN_HEAD16", "1") != "0"      # A/B knob: the two 1x1 heads of all levels as one launch per direction


@RPN_HEAD_REGISTRY.register()
class StandardRPNHead(nn.Module):
    """3x3 conv + ReLU, then 1x1 objectness (A) and 1x1 anchor deltas (4A).  The two 1x1 convs are
    evaluated as ONE 16-channel GEMM ([3 logits | 12 deltas | pad]) so the 256-channel feature map is
    read once; parameters stay separate (`conv`, `objectness_logits`, `anchor_deltas`)."""

    @configurable
    def __init__(self, *, in_channels, num_anchors, box_dim=4, conv_dims=(-1,)):
        super().__init__()
        assert len(conv_dims) == 1 and num_anchors == 3 and box_dim == 4, "hot path: A=3 anchors per location"
        out_channels = in_channels if conv_dims[0] == -1 else conv_dims[0]
        self.conv = Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.objectness_logits = Conv2d(out_channels, num_anchors, kernel_size=1, stride=1)
        self.anchor_deltas = Conv2d(out_channels, num_anchors * box_dim, kernel_size=1, stride=1)
        for layer in (self.conv, self.objectness_logits, self.anchor_deltas):
            nn.init.normal_(layer.weight, std=0.01)
            nn.init.constant_(layer.bias, 0)

    @classmethod
    def from_config(cls, cfg, input_shape):
        in_channels = [s.channels for s in input_shape]
        assert len(set(in_channels)) == 1
        ag = build_anchor_generator(cfg, input_shape)
        assert len(set(ag.num_anchors)) == 1
        return {"in_channels": in_channels[0], "num_anchors": ag.num_anchors[0], "box_dim": ag.box_dim,
                "conv_dims": cfg.MODEL.RPN.CONV_DIMS}

    def _shared_conv(self, features):
        """the shared 3x3 + ReLU over every level: one GEMM per direction for all levels when the maps allow it (functional.
        _WinoConv3x3Levels), else one convolution per level"""
        c = self.conv
        on = HF._WINO_LEVELS == "1" or (HF._WINO_LEVELS == "infer" and not (self.training and torch.is_grad_enabled()))
        if on and c.stride[0] == 1 and c.padding[0] == 1 and HF.conv3x3_levels_eligible(features, c.weight, 1, 1):
            return HF.conv3x3_levels(features, c.weight, c.bias, relu=True)
        return [c(x, relu=True) for x in features]

    def forward(self, features):
        wl, wd = self.objectness_logits.weight, self.anchor_deltas.weight
        if _FUSED_HEAD and HF.rpn_head16_eligible(features, wl, wd):
            # both 1x1 heads over every level: one HBM-bound launch per direction (csrc/rpn_head.hip)
            return HF.rpn_head16(self._shared_conv(features), wl, self.objectness_logits.bias, wd, self.anchor_deltas.bias)
        w16 = torch.cat([wl, wd, wl.new_zeros(1, wl.shape[1], 1, 1)], dim=0)
        b16 = torch.cat([self.objectness_logits.bias, self.anchor_deltas.bias, wl.new_zeros(1)])
        return [HF.conv2d(t, w16, b16, 1, 0) for t in self._shared_conv(features)]   # (B,16,H,W) CL each


def build_rpn_head(cfg, input_shape):
    return RPN_HEAD_REGISTRY.get(cfg.MODEL.RPN.HEAD_NAME)(cfg, input_shape)


@PROPOSAL_GENERATOR_REGISTRY.register()
class RPNWithIgnore(nn.Module):
    accepts_packed = True     # forward() also takes the ground truth pre-packed on the device (RCNN3D.prepack)
    @configurable
    def __init__(self, *, in_features, head, anchor_generator, anchor_thresholds, anchor_labels, batch_size_per_image,
                 positive_fraction, pre_nms_topk, post_nms_topk, nms_thresh=0.7, min_box_size=0.0, loss_weight=1.0,
                 box_reg_loss_type="smooth_l1", smooth_l1_beta=0.0, ignore_thresh=0.5, objectness_uncertainty="IoUness"):
        super().__init__()
        self.in_features = in_features
        self.rpn_head = head
        self.anchor_generator = anchor_generator
        self.anchor_thresholds, self.anchor_labels = list(anchor_thresholds), list(anchor_labels)
        self.batch_size_per_image = batch_size_per_image
        self.positive_fraction = positive_fraction
        self.pre_nms_topk = {True: pre_nms_topk[0], False: pre_nms_topk[1]}
        self.post_nms_topk = {True: post_nms_topk[0], False: post_nms_topk[1]}
        self.nms_thresh = nms_thresh
        self.min_box_size = float(min_box_size)
        self.loss_weight = loss_weight if isinstance(loss_weight, dict) else {"loss_rpn_cls": loss_weight, "loss_rpn_loc": loss_weight}
        self.ignore_thresh = ignore_thresh
        self.objectness_uncertainty = objectness_uncertainty
        if box_reg_loss_type != "smooth_l1" or smooth_l1_beta != 0.0:
            raise NotImplementedError("MI355X hot path: L1 anchor regression (BBOX_REG_LOSS_TYPE smooth_l1, SMOOTH_L1_BETA 0: Base.yaml / "
                                      "upstream defaults); the reference's IoUness losses only evaluate smooth_l1 as well (rpn.py:258-271)")
        if objectness_uncertainty.lower() not in ("iouness", "none"):
            # rpn.py:169-180 sends every other value down the same IoUness code path; only the two names that occur are accepted here
            raise NotImplementedError(f"MODEL.RPN.OBJECTNESS_UNCERTAINTY '{objectness_uncertainty}': 'IoUness' (Base.yaml:55) and 'none'")
        self.plain_objectness = objectness_uncertainty.lower() == "none"
        self.pending_logs = {}
        self.injected = None   # parity tests: dict with 'E' (B,A) exponential variates
        self.last = None

    @classmethod
    def from_config(cls, cfg, input_shape):
        in_features = cfg.MODEL.RPN.IN_FEATURES
        shapes = [input_shape[f] for f in in_features]
        return {
            "in_features": in_features, "min_box_size": cfg.MODEL.PROPOSAL_GENERATOR.MIN_SIZE,
            "nms_thresh": cfg.MODEL.RPN.NMS_THRESH, "batch_size_per_image": cfg.MODEL.RPN.BATCH_SIZE_PER_IMAGE,
            "positive_fraction": cfg.MODEL.RPN.POSITIVE_FRACTION,
            "loss_weight": {"loss_rpn_cls": cfg.MODEL.RPN.LOSS_WEIGHT,
                            "loss_rpn_loc": cfg.MODEL.RPN.BBOX_REG_LOSS_WEIGHT * cfg.MODEL.RPN.LOSS_WEIGHT},
            "box_reg_loss_type": cfg.MODEL.RPN.BBOX_REG_LOSS_TYPE, "smooth_l1_beta": cfg.MODEL.RPN.SMOOTH_L1_BETA,
            "pre_nms_topk": (cfg.MODEL.RPN.PRE_NMS_TOPK_TRAIN, cfg.MODEL.RPN.PRE_NMS_TOPK_TEST),
            "post_nms_topk": (cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN, cfg.MODEL.RPN.POST_NMS_TOPK_TEST),
            "anchor_generator": build_anchor_generator(cfg, shapes),
            "anchor_thresholds": cfg.MODEL.RPN.IOU_THRESHOLDS, "anchor_labels": cfg.MODEL.RPN.IOU_LABELS,
            "head": build_rpn_head(cfg, shapes), "ignore_thresh": cfg.MODEL.RPN.IGNORE_THRESHOLD,
            "objectness_uncertainty": cfg.MODEL.RPN.OBJECTNESS_UNCERTAINTY,
        }

    # ---- training targets (rpn.py:41-127) -----------------------------------------------------
    @torch.no_grad()
    def label_and_sample_anchors(self, anchors, targets):
        B, A = targets.B, anchors.shape[0]
        # injected Exp(1) variates (parity tests), else drawn inside the matching kernel (csrc/philox.h; round 6)
        E = self.injected["E"].to(anchors.device).float().contiguous() if (self.injected is not None and "E" in self.injected) else None
        if E is None and self.__dict__.get("_draw") is None:
            from ....kernels.glue import DrawState
            self.__dict__["_draw"] = DrawState()
        thr = self.anchor_thresholds
        m = det.rpn_match(anchors, targets.gt, targets.gt_off, E, (thr[0], thr[-1]), self.anchor_labels, True, draw=self.__dict__.get("_draw"), B=B)
        kpos = max(int(self.batch_size_per_image * self.positive_fraction), 1)
        # one launch for both draws: rows [0, B) = positive keys, [B, 2B) = negative keys (sorted => prefix = top-kpos)
        kv, ki = select.topk_rows(m["keys"], max(kpos, self.batch_size_per_image))
        pv, pi = kv[:B, :kpos].contiguous(), ki[:B, :kpos].contiguous()
        nv, ni = kv[B:, :self.batch_size_per_image].contiguous(), ki[B:, :self.batch_size_per_image].contiguous()
        labels, counts = det.rpn_finalize_labels(anchors, targets.gt_off, targets.ign, targets.ign_off, m, pv, pi, nv, ni,
                                                 self.batch_size_per_image, self.ignore_thresh)
        return labels, m["matched_idx"]

    def losses(self, levels, anchors, labels, matched_idx, targets):
        B = targets.B
        inv_norm = 1.0 / (self.batch_size_per_image * B)          # rpn.py:198
        names = ("rpn/cls", "rpn/loc")
        w = tuple(self.loss_weight.get(k, 1.0) for k in names)                     # rpn.py:203 (names never match => 1.0)
        vec, sums = HF.rpn_loss(levels, anchors, labels, matched_idx, targets.gt, targets.gt_off, inv_norm, w, self.plain_objectness)
        self.pending_logs = {"rpn": (sums, B)}
        return HF.LossDict({names[0]: vec[0], names[1]: vec[1]}, vectors=[(vec, names)])

    def flush_logs(self, storage):
        if "rpn" in self.pending_logs:
            sums, B = self.pending_logs.pop("rpn")
            s = sums.tolist()
            storage.put_scalar("rpn/num_pos_anchors", s[2] / B)
            storage.put_scalar("rpn/num_neg_anchors", s[3] / B)
            if not self.plain_objectness:        # logged by the IoUness loss only (rpn.py:252-255)
                storage.put_scalar("rpn/conf_pos_anchors", s[4] / max(s[2], 1.0))
                n_other = B * self._num_anchors - s[2]
                storage.put_scalar("rpn/conf_neg_anchors", s[5] / max(n_other, 1.0))

    # ---- proposals (detectron2 predict_proposals / find_top_rpn_proposals) ---------------------
    @torch.no_grad()
    def predict_proposals(self, levels, anchors, hw_list, image_hw):
        training = self.training
        pre, post = self.pre_nms_topk[training], self.post_nms_topk[training]
        lv = [t.detach().permute(0, 2, 3, 1) for t in levels]
        pack = det.LevelPack(lv)
        B = pack.B
        logits = det.rpn_gather_logits(pack)                                    # (B, A)
        # every level gets `kmax` slots (top-k pads with -inf / -1 when a level has fewer anchors), so the
        # (B, L*kmax) block is L*B equally sized, score-sorted NMS problems: ONE decode, ONE NMS launch
        L = len(hw_list)
        kmax = min(pre, max(H * W * 3 for H, W in hw_list))
        seg_n = [H * W * 3 for H, W in hw_list]
        seg_off = [sum(seg_n[:l]) for l in range(L)]
        scores, idx = select.topk_segments(logits, seg_off, seg_n, kmax)         # all levels, one launch
        scores, idx = scores.view(B, L * kmax), idx.view(B, L * kmax)
        key = (L, kmax, str(idx.device))
        if getattr(self, "_slot_key", None) != key:
            self._slot_level = torch.arange(L, dtype=torch.int32, device=idx.device).repeat_interleave(kmax).contiguous()
            self._slot_key = key
        boxes, valid = det.rpn_decode(pack, self._slot_level, idx, anchors, image_hw, self.min_box_size)
        keep = select.nms_sorted(boxes.view(B * L, kmax, 4), self.nms_thresh, None, valid.view(B * L, kmax)).view(B, L * kmax)
        masked = det.rpn_mask_scores(scores.contiguous(), keep.contiguous())     # suppressed / invalid candidates -> -inf
        top_v, top_i = select.topk_rows(masked, post)                             # keep[:post_nms_topk], score order
        prop, count = det.rpn_collect(boxes, top_v, top_i)                        # boxes of the real entries, zeros behind; #real
        if getattr(self, "keep_candidates", False):      # parity tests: the pre-NMS candidate lists behind the proposal set
            self.last_candidates = {"boxes": boxes, "scores": scores, "keep": keep, "valid": valid.view(B, L * kmax), "slots_per_level": kmax}
        return prop, top_v, count

    def forward(self, images, features, gt_instances=None, targets=None):
        """Reference contract (rpn.py / detectron2 RPN.forward): `gt_instances` = list[Instances] with gt_boxes / gt_classes
        (class -1 = ignore region).  `targets` = the same ground truth already packed by RCNN3D.prepack (this package's
        own meta-architecture passes it so the ROI heads reuse the device copy); either form works."""
        if targets is None and gt_instances is not None:
            from ..targets import pack_instances_cached
            targets = pack_instances_cached(gt_instances, images.image_sizes, device=images.tensor.device)
        feats = [features[f] for f in self.in_features]
        hw_list = [(f.shape[2], f.shape[3]) for f in feats]
        anchors = self.anchor_generator.grid(hw_list, feats[0].device)
        self._num_anchors = anchors.shape[0]
        levels = self.rpn_head(feats)
        if self.training:
            assert targets is not None, "RPN requires ground truth in training!"
            self.__dict__["_last_hw_list"] = hw_list
            hook = self.__dict__.get("_label_split")
            if hook is not None:
                # a step being captured in pipelined form (solver/graphed.py, OMNI_PIPE_LABELS): anchor labelling + sampling read only
                # the anchors and the ground truth, so they were captured as a graph of their own that replays on the idle
                # weight-gradient stream beside the bottom-up; the critical-path graph is cut HERE and its second half starts behind both
                labels, matched_idx = hook()
            else:
                with HF.forked():        # (a parallel branch of the captured step, joined behind the proposals; eager: a no-op)
                    labels, matched_idx = self.label_and_sample_anchors(anchors, targets)
            losses = self.losses(levels, anchors, labels, matched_idx, targets)
            self.last_labels = labels
        else:
            losses = {}
        image_hw = targets.image_hw if targets is not None else torch.tensor(
            [list(s) for s in images.image_sizes], dtype=torch.int32, device=anchors.device)
        prop, scores, count = self.predict_proposals(levels, anchors, hw_list, image_hw)
        HF.join_branch()
        self.last = {"boxes": prop, "scores": scores, "count": count}
        if self.injected is not None and "proposals" in self.injected:
            # stage-wise parity tests: the second stage runs on a given proposal list (list of (n_i, 4) boxes in score
            # order), so that a last-bit flip in the first stage's ranking does not re-deal the sampling variates
            given = self.injected["proposals"]
            prop = torch.zeros_like(prop)
            for n, b in enumerate(given):
                prop[n, :b.shape[0]] = b.to(prop.device)
            count = torch.tensor([b.shape[0] for b in given], dtype=torch.int32, device=prop.device)
        return _PackedProposals(prop, scores, count, images.image_sizes), losses


class _PackedProposals(list):
    """list[Instances] view of the packed (B, post_nms_topk, 4) proposal tensor.  Materialising the
    per-image Instances needs the counts on the host (one sync); the ROI heads of this package read the
    packed tensors directly and never trigger it."""

    def __init__(self, boxes, scores, count, image_sizes):
        super().__init__()
        self.boxes, self.scores, self.count, self.image_sizes = boxes, scores, count, image_sizes
        self._done = False

    def _materialise(self):
        if not self._done:
            self._done = True
            for n, c in enumerate(self.count.tolist()):
                inst = Instances(tuple(self.image_sizes[n]))
                inst.proposal_boxes = Boxes(self.boxes[n, :c])
                inst.objectness_logits = self.scores[n, :c]
                super().append(inst)

    def __len__(self):
        return len(self.image_sizes)

    def __getitem__(self, i):
        self._materialise()
        return super().__getitem__(i)

    def __iter__(self):
        self._materialise()
        return super().__iter__()


def build_proposal_generator(cfg, input_shape):
    name = cfg.MODEL.PROPOSAL_GENERATOR.NAME
    if name == "PrecomputedProposals":
        return None
    return PROPOSAL_GENERATOR_REGISTRY.get(name)(cfg, input_shape)
