"""`ROIHeads3D` (ROI_HEADS_REGISTRY) on the HIP kernels.

Mirrors /root/reference/cubercnn/modeling/roi_heads/roi_heads.py: constructor / from_config keys
(:42-204), `forward(images, features, proposals, Ks, im_scales_ratio, targets)` (:207-246),
`label_and_sample_proposals` (:862-929), `_forward_box` (:249-293), `_forward_cube` (:326-824) with
the same loss names, loss weights and logged scalars -- for every MODEL.ROI_CUBE_HEAD switch the
reference evaluates (configs/Base.yaml: disentangled + chamfer + joint losses, virtual depth, allocentric
6D pose, dimension priors, confidence; and Z_TYPE / CLUSTER_BINS / POSE_TYPE / DISENTANGLED_LOSS /
SCALE_ROI_BOXES / TRAIN_ON_PRED_BOXES variants).

Data layout: the sampled ROIs of a batch are a fixed-shape (B, batch_size_per_image) block with the
foreground first in every row (the sampler's order), so the 2D box head runs on B*512 rows and the
cube head on the first `fg_cap` = int(512 * positive_fraction) slots of every image; padding /
background slots carry class -2 / K and are skipped inside the loss kernels.  No per-image Python
loops, boolean-mask gathers or host syncs."""
import torch
from torch import nn

from .... import functional as HF
from ....d2.config import configurable
from ....d2.layers import ShapeSpec
from ....d2.structures import Boxes, Instances
from ....kernels import det
from ..layers import FlattenLinear, Linear
from ..registries import ROI_BOX_HEAD_REGISTRY, ROI_HEADS_REGISTRY
from .cube_head import build_cube_head
from .fast_rcnn import FastRCNNOutputs
from ..targets import MAX_GT_PER_IMAGE


def build_roi_heads(cfg, input_shape, priors=None):
    return ROI_HEADS_REGISTRY.get(cfg.MODEL.ROI_HEADS.NAME)(cfg, input_shape, priors=priors)


@ROI_BOX_HEAD_REGISTRY.register()
class FastRCNNConvFCHead(nn.Module):
    """detectron2 FastRCNNConvFCHead with NUM_CONV 0: flatten -> fc1 -> ReLU [-> fc2 -> ReLU ...] (state-dict names `fc1` .. `fcN`;
    configs/Base.yaml:67-70 uses two)."""

    def __init__(self, cfg, input_shape):
        super().__init__()
        b = cfg.MODEL.ROI_BOX_HEAD
        if b.NUM_CONV != 0 or b.NUM_FC < 1 or b.NORM != "":
            raise NotImplementedError("MI355X hot path: FastRCNNConvFCHead with NUM_CONV 0, NUM_FC >= 1, no norm (Base.yaml:67-70 uses 2 FCs)")
        self.num_fc = b.NUM_FC
        self.fc1 = FlattenLinear(input_shape.channels, input_shape.height, b.FC_DIM)
        for k in range(2, self.num_fc + 1):
            fc = Linear(b.FC_DIM, b.FC_DIM)
            nn.init.kaiming_uniform_(fc.weight, a=1)
            nn.init.constant_(fc.bias, 0)
            setattr(self, f"fc{k}", fc)
        self._fc_dim = b.FC_DIM

    @property
    def output_shape(self):
        return ShapeSpec(channels=self._fc_dim)

    def forward(self, x):
        x = self.fc1(x, relu=True)
        for k in range(2, self.num_fc + 1):
            x = getattr(self, f"fc{k}")(x, relu=True)
        return x


class ROIPooler(nn.Module):
    """detectron2 ROIPooler (ROIAlignV2, canonical size 224 / level 4) over packed ROI tensors."""

    def __init__(self, output_size, scales, sampling_ratio, pooler_type, canonical_box_size=224, canonical_level=4):
        super().__init__()
        import math
        # POOLER_TYPE is passed through by the reference (roi_heads.py:166-171).  "ROIAlignV2" is every released configuration;
        # "ROIAlign" (no half-pixel shift) and "ROIPool" (torchvision roi_pool; it has no sampling ratio) run on the general kernels since
        # round 6; "ROIAlignRotated" takes RotatedBoxes (five numbers per box), which this model's proposals are not.
        if pooler_type not in ("ROIAlignV2", "ROIAlign", "ROIPool") or (sampling_ratio != 0 and pooler_type != "ROIPool"):
            raise NotImplementedError(f"MI355X hot path: POOLER_TYPE ROIAlignV2 / ROIAlign with adaptive sampling (POOLER_SAMPLING_RATIO 0) or "
                                      f"ROIPool; got {pooler_type!r} / {sampling_ratio}")
        self.aligned = pooler_type == "ROIAlignV2"
        self.max_pool = pooler_type == "ROIPool"
        self.output_size = output_size if isinstance(output_size, int) else output_size[0]
        self.scales = tuple(scales)
        self.min_level = int(round(-math.log2(scales[0])))
        self.max_level = int(round(-math.log2(scales[-1])))
        self.canonical_box_size, self.canonical_level = canonical_box_size, canonical_level

    def forward(self, feats, rois, batch_idx):
        levels = det.roi_levels(rois, self.min_level, self.max_level, float(self.canonical_box_size), self.canonical_level)
        if self.max_pool:
            return HF.roi_pool(feats, self.scales, rois, batch_idx, levels, self.output_size)
        if not self.aligned:
            return HF.roi_align_legacy(feats, self.scales, rois, batch_idx, levels, self.output_size)
        return HF.roi_align(feats, self.scales, rois, batch_idx, levels, self.output_size)

    def same_as(self, other):
        """True when one shared pooling pass can serve both poolers (the aligned form only has the shared kernels)"""
        return self.aligned and other.aligned and (self.output_size, self.scales, self.canonical_box_size, self.canonical_level) == \
            (other.output_size, other.scales, other.canonical_box_size, other.canonical_level)

    def forward_shared(self, feats, rois, batch_idx, per_image, first):
        """-> (pooled features of all ROIs, pooled features of the first `first` ROIs of every image), one kernel pass."""
        levels = det.roi_levels(rois, self.min_level, self.max_level, float(self.canonical_box_size), self.canonical_level)
        return HF.roi_align_shared(feats, self.scales, rois, batch_idx, levels, self.output_size, per_image, first)


def _pack_proposal_list(proposals, image_sizes, device):
    """list[Instances] (proposal_boxes[, objectness_logits], score order) -> the packed (B, P, 4) + count form"""
    from ..proposal_generator.rpn import _PackedProposals
    P = max([len(p) for p in proposals] + [1])
    boxes = torch.zeros((len(proposals), P, 4), dtype=torch.float32, device=device)
    scores = torch.full((len(proposals), P), float("-inf"), dtype=torch.float32, device=device)
    for n, p in enumerate(proposals):
        k = len(p)
        if k:
            boxes[n, :k] = p.proposal_boxes.tensor.to(device)
            if p.has("objectness_logits"):
                scores[n, :k] = p.objectness_logits.to(device)
    count = torch.tensor([len(p) for p in proposals], dtype=torch.int32, device=device)
    return _PackedProposals(boxes, scores, count, image_sizes)


@ROI_HEADS_REGISTRY.register()
class ROIHeads3D(nn.Module):
    accepts_packed = True     # forward() also takes the ground truth pre-packed on the device (RCNN3D.prepack)
    replayable_inference = True    # its eval pass is inference.roi_heads_inference_device + one host sync (meta_arch/infer_replay.py)
    pool_cut = None           # solver/graphed.py: callable applied to the pooled (box, cube) features = a backward stage boundary
    @configurable
    def __init__(self, *, num_classes, batch_size_per_image, positive_fraction, proposal_iou_threshold, proposal_append_gt,
                 box_in_features, box_pooler, box_head, box_predictor, ignore_thresh, cube_head, cube_pooler, loss_w_3d,
                 loss_w_xy, loss_w_z, loss_w_dims, loss_w_pose, loss_w_joint, use_confidence, inverse_z_weight, z_type,
                 pose_type, cluster_bins, priors=None, dims_priors_enabled=None, dims_priors_func=None, disentangled_loss=None,
                 virtual_depth=None, virtual_focal=None, test_scale=None, allocentric_pose=None, chamfer_pose=None,
                 scale_roi_boxes=None, train_on_pred_boxes=False):
        super().__init__()
        self.num_classes = num_classes
        self.batch_size_per_image = batch_size_per_image
        self.positive_fraction = positive_fraction
        self.proposal_iou_threshold = proposal_iou_threshold
        self.proposal_append_gt = proposal_append_gt
        self.in_features = self.box_in_features = box_in_features
        self.box_pooler, self.box_head, self.box_predictor = box_pooler, box_head, box_predictor
        self.ignore_thresh = ignore_thresh
        self.loss_w_3d, self.loss_w_xy, self.loss_w_z = loss_w_3d, loss_w_xy, loss_w_z
        self.loss_w_dims, self.loss_w_pose, self.loss_w_joint = loss_w_dims, loss_w_pose, loss_w_joint
        self.use_confidence, self.virtual_focal, self.test_scale = use_confidence, virtual_focal, test_scale
        self.cluster_bins, self.z_type = cluster_bins, z_type
        self.scale_roi_boxes = float(scale_roi_boxes or 0.0)
        self.train_on_pred_boxes = bool(train_on_pred_boxes)
        if loss_w_3d <= 0:
            raise NotImplementedError("MI355X hot path: LOSS_W_3D <= 0 (no cube head) is not built; the reference's training branch "
                                      "fails there as well (roi_heads.py:219-225 returns an unassigned `instances_3d`)")
        # head parameterisation + loss switches for csrc/cube_head.hip (bit layout: include/omni3d_hip.h)
        self.cube_mode = det.cube_mode(z_type, dims_priors_enabled, dims_priors_func, pose_type, allocentric_pose, virtual_depth,
                                       chamfer_pose, inverse_z_weight, use_confidence > 0, loss_w_joint > 0,
                                       disentangled=bool(disentangled_loss))
        self.cube_head, self.cube_pooler = cube_head, cube_pooler
        if priors is not None:
            self.priors_dims_per_cat = nn.Parameter(torch.FloatTensor(priors["priors_dims_per_cat"]).unsqueeze(0))
        else:
            self.priors_dims_per_cat = nn.Parameter(torch.ones(1, num_classes, 2, 3))
        # depth clusters over the 2D scale (roi_heads.py:122-143); priors['priors_bins'] = [(name, scales, [[z mean, z std]])]
        bins_table = priors.get("priors_bins") if priors is not None else None
        if cluster_bins > 1 and bins_table:
            self.priors_z_scales = nn.Parameter(torch.stack([torch.FloatTensor(p[1]) for p in bins_table]))
        else:
            self.priors_z_scales = nn.Parameter(torch.ones(num_classes, cluster_bins))
        if z_type == "clusters":
            assert cluster_bins > 1, "To use z_type of priors, there must be more than 1 cluster bin"
            if bins_table:
                self.priors_z_stats = nn.Parameter(torch.cat([torch.FloatTensor(p[2]).unsqueeze(0) for p in bins_table]))
            else:
                self.priors_z_stats = nn.Parameter(torch.ones(num_classes, cluster_bins, 2).float())
        self.pending_logs = {}
        self.injected = None     # parity tests: {'E': (B, 2048) exponential variates}
        self.fg_cap = int(batch_size_per_image * positive_fraction)

    @classmethod
    def from_config(cls, cfg, input_shape, priors=None):
        in_features = cfg.MODEL.ROI_HEADS.IN_FEATURES
        scales = tuple(1.0 / input_shape[k].stride for k in in_features)
        channels = [input_shape[f].channels for f in in_features]
        assert len(set(channels)) == 1, channels
        bres = cfg.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION
        box_head = ROI_BOX_HEAD_REGISTRY.get(cfg.MODEL.ROI_BOX_HEAD.NAME)(cfg, ShapeSpec(channels=channels[0], height=bres, width=bres))
        c = cfg.MODEL.ROI_CUBE_HEAD
        cres = c.POOLER_RESOLUTION
        assert len(cfg.MODEL.ROI_HEADS.IOU_THRESHOLDS) == 1
        return {
            "num_classes": cfg.MODEL.ROI_HEADS.NUM_CLASSES, "batch_size_per_image": cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE,
            "positive_fraction": cfg.MODEL.ROI_HEADS.POSITIVE_FRACTION, "proposal_iou_threshold": cfg.MODEL.ROI_HEADS.IOU_THRESHOLDS[0],
            "proposal_append_gt": cfg.MODEL.ROI_HEADS.PROPOSAL_APPEND_GT, "train_on_pred_boxes": cfg.MODEL.ROI_BOX_HEAD.TRAIN_ON_PRED_BOXES,
            "box_in_features": in_features,
            "box_pooler": ROIPooler(bres, scales, cfg.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO, cfg.MODEL.ROI_BOX_HEAD.POOLER_TYPE),
            "box_head": box_head, "box_predictor": FastRCNNOutputs(cfg, box_head.output_shape),
            "cube_pooler": ROIPooler(cres, scales, c.POOLER_SAMPLING_RATIO, c.POOLER_TYPE),
            "cube_head": build_cube_head(cfg, ShapeSpec(channels=channels[0], width=cres, height=cres)),
            "use_confidence": c.USE_CONFIDENCE, "inverse_z_weight": c.INVERSE_Z_WEIGHT, "loss_w_3d": c.LOSS_W_3D,
            "loss_w_xy": c.LOSS_W_XY, "loss_w_z": c.LOSS_W_Z, "loss_w_dims": c.LOSS_W_DIMS, "loss_w_pose": c.LOSS_W_POSE,
            "loss_w_joint": c.LOSS_W_JOINT, "z_type": c.Z_TYPE, "pose_type": c.POSE_TYPE,
            "dims_priors_enabled": c.DIMS_PRIORS_ENABLED, "dims_priors_func": c.DIMS_PRIORS_FUNC,
            "disentangled_loss": c.DISENTANGLED_LOSS, "virtual_depth": c.VIRTUAL_DEPTH, "virtual_focal": c.VIRTUAL_FOCAL,
            "test_scale": cfg.INPUT.MIN_SIZE_TEST, "chamfer_pose": c.CHAMFER_POSE, "allocentric_pose": c.ALLOCENTRIC_POSE,
            "cluster_bins": c.CLUSTER_BINS, "ignore_thresh": cfg.MODEL.RPN.IGNORE_THRESHOLD, "scale_roi_boxes": c.SCALE_ROI_BOXES,
            "priors": priors,
        }

    # ---- roi_heads.py:862-929 --------------------------------------------------------------------
    @torch.no_grad()
    def label_and_sample_proposals(self, proposals, targets):
        boxes, count = proposals.boxes, proposals.count
        B = boxes.shape[0]
        if boxes.shape[1] + MAX_GT_PER_IMAGE > det.ROI_MAXC and self.proposal_append_gt:
            raise ValueError(f"{boxes.shape[1]} proposals + up to {MAX_GT_PER_IMAGE} appended GT boxes exceed the sampler's "
                             f"capacity of {det.ROI_MAXC} candidates per image (csrc/rpn_roi.hip ROI_MAXC)")
        # the sampling randomness: injected Exp(1) variates (parity tests), else drawn INSIDE the sampling kernel (round 6: no
        # `exponential_()` launch and none of the generator bookkeeping torch wraps around it under graph capture)
        E = self.injected["E"].to(boxes.device).float().contiguous() if (self.injected is not None and "E" in self.injected) else None
        if E is None and self.__dict__.get("_draw") is None:
            from ....kernels.glue import DrawState
            self.__dict__["_draw"] = DrawState()
        out = det.roi_sample(boxes, count, targets.gt, targets.gt_cls, targets.gt_off, targets.ign, targets.ign_off, E,
                             self.proposal_iou_threshold, self.ignore_thresh, self.num_classes, self.batch_size_per_image,
                             self.positive_fraction, self.proposal_append_gt, draw=self.__dict__.get("_draw"), first=self.fg_cap)
        self.pending_logs["roi_counts"] = out[4]
        self.last_sampled_boxes, self.last_sampled_classes = out[0], out[1]    # (B, S, 4) / (B, S); read by the parity tests
        # what the kernel prepared for the losses: ground-truth rows with the background marker clamped to row 0, and the contiguous
        # prefix (boxes, classes, rows) of the cube head's slots -- `sgt.clamp(min=0)` and three slice copies before
        self.__dict__["_gt_rows_memo"] = (out[2], out[5])
        self.__dict__["_cube_prefix"] = (out[0], out[6])
        return out[:5]

    def forward(self, images, features, proposals, Ks, im_scales_ratio, targets=None, packed=None):
        """Reference contract (roi_heads.py:207): proposals = list[Instances] (proposal_boxes), Ks = per-image 3x3 intrinsics
        of the original image, im_scales_ratio = original height / network height, targets = list[Instances] ground truth.
        `packed` = all of that already on the device (RCNN3D.prepack); this package's RCNN3D passes it and None for the
        rest.  Proposals may be this package's packed list or any list[Instances] (e.g. from a reference RPN)."""
        feats = [features[f] for f in self.in_features]
        if packed is None:
            from ..targets import pack_instances_cached
            assert Ks is not None and im_scales_ratio is not None, "ROIHeads3D needs Ks and im_scales_ratio (roi_heads.py:207)"
            packed = pack_instances_cached(targets if targets is not None else [None] * len(images.image_sizes), images.image_sizes,
                                           Ks, im_scales_ratio, self.virtual_focal, device=images.tensor.device)
        if (not self.training and isinstance(proposals, list) and len(proposals) > 0
                and not any(isinstance(p, Instances) for p in proposals)):
            # oracle 2D boxes (rcnn3d.py:98-101, roi_heads.py:228-240): a list of {'gt_bbox2D', 'gt_classes'} bypasses RPN and box head
            from .inference import roi_heads_oracle2d
            return roi_heads_oracle2d(self, images, feats, proposals, packed), {}
        if not hasattr(proposals, "boxes"):
            proposals = _pack_proposal_list(proposals, images.image_sizes, images.tensor.device)
        if self.training:
            assert packed.num_gt >= 0
            sboxes, scls, sgt, siou, counts = self.label_and_sample_proposals(proposals, packed)
            x_box = x_cube = None
            own_cube_rois = self.train_on_pred_boxes or self.scale_roi_boxes > 0
            if self.box_pooler.same_as(self.cube_pooler) and not own_cube_rois:   # every released config: pool once for both heads
                B, S = scls.shape
                x_box, x_cube = self.box_pooler.forward_shared(feats, sboxes.reshape(B * S, 4), self._batch_index(B, S, sboxes.device),
                                                               S, self.fg_cap)
                if self.pool_cut is not None and torch.is_grad_enabled():      # solver/graphed.py: backward stage boundary
                    x_box, x_cube = self.pool_cut((x_box, x_cube))
            losses, cube_boxes = self._forward_box_train(feats, sboxes, scls, sgt, packed, x_box)
            losses.update(self._forward_cube_train(feats, cube_boxes, scls, sgt, packed, x_cube))
            return [], losses
        from .inference import roi_heads_inference
        return roi_heads_inference(self, images, feats, proposals, packed), {}

    def _gt_rows(self, sgt):
        """matched ground-truth row per sampled ROI with the background marker (-1) clamped to row 0, computed ONCE per step for the box
        and the cube losses (two launches less)"""
        memo = self.__dict__.get("_gt_rows_memo")
        if memo is None or memo[0] is not sgt:
            memo = self.__dict__["_gt_rows_memo"] = (sgt, sgt.clamp(min=0))
        return memo[1]

    def _batch_index(self, B, per_image, device):
        cache = self.__dict__.setdefault("_bidx_cache", {})
        key = (B, per_image, str(device))
        if key not in cache:
            cache[key] = torch.arange(B, dtype=torch.int32, device=device).repeat_interleave(per_image).contiguous()
        return cache[key]

    # ---- roi_heads.py:249-293 + fast_rcnn.py:145-194 ------------------------------------------------
    def _forward_box_train(self, feats, sboxes, scls, sgt, packed, x=None):
        B, S = scls.shape
        rois = sboxes.reshape(B * S, 4)
        if x is None:
            x = self.box_pooler(feats, rois, self._batch_index(B, S, rois.device))
        pred = self.box_predictor(self.box_head(x))
        losses = self.box_predictor.losses(pred, scls.reshape(-1), rois, packed, self._gt_rows(sgt).reshape(-1))
        if self.train_on_pred_boxes:      # roi_heads.py:283-289: the 3D head trains on the (detached) 2D predictions of the GT classes
            with torch.no_grad():
                sboxes = det.box_decode_gt_class(pred.detach().contiguous(), self.num_classes, scls.reshape(-1).contiguous(), rois.contiguous(),
                                                 self.box_predictor.box2box_weights).view(B, S, 4)
        return losses, sboxes

    def clusters(self):
        """None, or (bins, 2D-scale priors (K, bins), depth priors (K, bins, 2) | None) for the cube kernels"""
        if self.cluster_bins <= 1:
            return None
        stats = self.priors_z_stats.detach().contiguous() if self.z_type == "clusters" else None
        return (self.cluster_bins, self.priors_z_scales.detach().contiguous(), stats)

    def scale_proposals(self, rois):
        """roi_heads.py:307-324: zoom the boxes the cube features are pooled from; both extents use the WIDTH, as the reference does"""
        if not self.scale_roi_boxes > 0:
            return rois
        cx, cy = (rois[:, 0] + rois[:, 2]) / 2, (rois[:, 1] + rois[:, 3]) / 2
        half = 0.5 * (rois[:, 2] - rois[:, 0]) * self.scale_roi_boxes
        return torch.stack([cx - half, cy - half, cx + half, cy + half], dim=1).contiguous()

    # ---- roi_heads.py:326-768 (training path) ----------------------------------------------------
    def _forward_cube_train(self, feats, sboxes, scls, sgt, packed, x=None):
        B, S = scls.shape
        Fc = self.fg_cap
        pre = self.__dict__.get("_cube_prefix")
        if pre is not None and pre[0] is sboxes and pre[1][0] is not None and pre[1][0].shape[1] == min(Fc, S):
            rois, cls, gt_row = pre[1][0].reshape(-1, 4), pre[1][1].reshape(-1), pre[1][2].reshape(-1)     # written by the sampling kernel
        else:      # (boxes that are not the sampler's own: TRAIN_ON_PRED_BOXES, a caller's sample)
            rois = sboxes[:, :Fc].reshape(B * Fc, 4).contiguous()
            cls = scls[:, :Fc].reshape(-1).contiguous()
            gt_row = self._gt_rows(sgt)[:, :Fc].reshape(-1).contiguous()
        bidx = self._batch_index(B, Fc, rois.device)
        if x is None:
            x = self.cube_pooler(feats, self.scale_proposals(rois), bidx)
        head = self.cube_head(x)
        priors = self.priors_dims_per_cat.detach().reshape(self.num_classes, 2, 3).contiguous()
        w3 = self.loss_w_3d
        # what each reduced term is multiplied by on the way into the loss dict; 0 = not a loss in this configuration
        coef = (self.loss_w_dims * w3 if self.loss_w_dims > 0 else 0.0,        # roi_heads.py:745-748
                self.loss_w_xy * w3, self.loss_w_z * w3, self.loss_w_pose * w3,
                self.loss_w_joint * w3 if self.loss_w_joint > 0 else 0.0,      # roi_heads.py:766-768
                float(self.use_confidence) if self.use_confidence > 0 else 0.0)   # roi_heads.py:721-740
        vec, red = HF.cube_loss(head, self.num_classes, rois, cls, bidx, packed.Ks, packed.v2r, priors, packed.gt3d,
                                packed.gtpose, gt_row, (self.loss_w_dims, self.loss_w_pose, self.loss_w_xy, self.loss_w_z, self.loss_w_joint),
                                self.cube_mode, coef, self.clusters())
        self.pending_logs["cube"] = red
        order = ("loss_dims", "loss_xy", "loss_z", "loss_pose", "loss_joint", "uncert")
        keep = [k for k, name in enumerate(order) if coef[k] != 0.0 or name in ("loss_xy", "loss_z", "loss_pose")]
        names = tuple("Cube/" + order[k] for k in keep)
        return HF.LossDict({n: vec[k] for n, k in zip(names, keep)}, vectors=[(vec, names)])

    def flush_logs(self, storage):
        self.box_predictor.flush_logs(storage)
        if "roi_counts" in self.pending_logs:
            c = self.pending_logs.pop("roi_counts").float().mean(0).tolist()
            storage.put_scalar("roi_head/num_fg_samples", c[0])
            storage.put_scalar("roi_head/num_bg_samples", c[1])
        if "cube" in self.pending_logs:
            r = self.pending_logs.pop("cube").tolist()
            for name, k in (("z_error", 13), ("dims_error", 14), ("xy_error", 15), ("z_close", 16), ("conf", 17)):
                if name == "conf" and not self.use_confidence > 0:
                    continue
                storage.put_scalar("Cube/" + name, r[k], smoothing_hint=False)
            storage.put_scalar("Cube/total_3D_loss", self.loss_w_3d * r[12], smoothing_hint=False)
