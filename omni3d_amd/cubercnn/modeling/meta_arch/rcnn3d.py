"""`RCNN3D` (META_ARCH_REGISTRY), `build_model`, `build_backbone`.

Mirrors /root/reference/cubercnn/modeling/meta_arch/rcnn3d.py (:25-112, :247-272): constructor via
`from_config(cfg, priors)`, `forward(batched_inputs)` returning the loss dict in training and
`[{"instances": Instances}]` in eval, `.device`, `preprocess_image`.  The image batch is normalised,
padded and converted to NHWC (C padded 3 -> 4) by one kernel; ground truth is packed once per step."""
import torch
from torch import nn

from ....d2.config import configurable
from ....d2.events import get_event_storage, has_event_storage
from ....d2.layers import ShapeSpec
from ....d2.structures import ImageList
from .... import functional as HF
from ....kernels import bnpool
from ..proposal_generator import build_proposal_generator
from ..registries import BACKBONE_REGISTRY, META_ARCH_REGISTRY
from ..roi_heads import build_roi_heads
from ..targets import pack_targets


_EVAL_F22 = __import__("os").environ.get("OMNI_EVAL_F22", "1") != "0"       # A/B knob


@META_ARCH_REGISTRY.register()
class RCNN3D(nn.Module):
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
        self._mean, self._std = [float(v) for v in pixel_mean], [float(v) for v in pixel_std]

    @classmethod
    def from_config(cls, cfg, priors=None):
        backbone = build_backbone(cfg, priors=priors)
        return {
            "backbone": backbone,
            "proposal_generator": build_proposal_generator(cfg, backbone.output_shape()),
            "roi_heads": build_roi_heads(cfg, backbone.output_shape(), priors=priors),
            "input_format": cfg.INPUT.FORMAT, "vis_period": cfg.VIS_PERIOD,
            "pixel_mean": cfg.MODEL.PIXEL_MEAN, "pixel_std": cfg.MODEL.PIXEL_STD,
        }

    @property
    def device(self):
        return self.pixel_mean.device

    def preprocess_image(self, batched_inputs, slot_hw=None, hw_dev=None):
        """slot_hw: device (B, 2) int32 -- the images sit in equal-size slots (the size-bucketed replay of solver/autoreplay.py) and
        their real sizes are device data; the padding mask is then applied by the kernel, not by host-side slicing"""
        imgs = [x["image"] for x in batched_inputs]
        sizes = [(im.shape[-2], im.shape[-1]) for im in imgs]
        # equal-size images already on the device are read through their own pointers: no `torch.stack` copy (round 6)
        direct = (len(set(sizes)) == 1 and all(im.device == self.device for im in imgs))
        if slot_hw is not None:
            assert len(set(sizes)) == 1
            x = bnpool.preprocess_list(imgs, self._mean, self._std, self.backbone.size_divisibility, image_hw=slot_hw) if direct else None
            if x is None:
                batch = torch.stack(imgs).to(self.device, non_blocking=True)
                x = bnpool.preprocess(batch, self._mean, self._std, self.backbone.size_divisibility, image_hw=slot_hw)
            return ImageList(x, sizes)
        if direct:
            x = bnpool.preprocess_list(imgs, self._mean, self._std, self.backbone.size_divisibility)
            if x is not None:
                return ImageList(x, sizes)
        if len(set(sizes)) != 1 and hw_dev is not None and self.device.type == "cuda":
            # ragged batch, sizes already on the device (packed.image_hw): every image is copied into the corner of an UNINITIALISED
            # device slot and the kernel masks the rest (round 6).  The host-side form below zero-fills and slice-copies a 14 MB CPU
            # tensor with torch's intra-op thread pool on every iteration -- the op that woke 128 threads under a 16-CPU cgroup quota
            # (omni3d_amd.respect_cpu_quota) -- and re-zeroes the padding with 2 launches per image afterwards.
            H, W = max(s_[0] for s_ in sizes), max(s_[1] for s_ in sizes)
            slots = torch.empty((len(imgs), 3, H, W), dtype=torch.uint8, device=self.device)
            for n, im in enumerate(imgs):
                slots[n, :, : im.shape[-2], : im.shape[-1]].copy_(im, non_blocking=True)
            x = bnpool.preprocess(slots, self._mean, self._std, self.backbone.size_divisibility, image_hw=hw_dev)
            return ImageList(x, sizes)
        if len(set(sizes)) == 1:
            batch = torch.stack(imgs).to(self.device, non_blocking=True)
        else:   # ragged batch: pad the uint8 images on the host first (zero padding is re-zeroed after normalisation)
            H, W = max(s[0] for s in sizes), max(s[1] for s in sizes)
            batch = torch.zeros((len(imgs), 3, H, W), dtype=torch.uint8)
            for b, im in zip(batch, imgs):
                b[:, : im.shape[-2], : im.shape[-1]] = im
            batch = batch.to(self.device, non_blocking=True)
        x = bnpool.preprocess(batch, self._mean, self._std, self.backbone.size_divisibility)
        if len(set(sizes)) != 1:
            for n, (h, w) in enumerate(sizes):   # re-zero the region that was host padding
                x[n, :, h:, :] = 0
                x[n, :, :, w:] = 0
        return ImageList(x, sizes)

    def prepack(self, batched_inputs):
        """Pack the ground truth / intrinsics of a batch once and keep them on the device (benchmarks
        pre-stage their synthetic batches with this; the training loop calls it implicitly)."""
        sizes = [(x["image"].shape[-2], x["image"].shape[-1]) for x in batched_inputs]
        vf = getattr(self.roi_heads, "virtual_focal", 512.0)
        return pack_targets(batched_inputs, sizes, vf, with_gt=True).to(self.device)

    def forward(self, batched_inputs, packed=None):
        auto = self.__dict__.get("_omni_auto") if self.training else None
        if auto is not None and packed is None and not auto.busy:
            # the drop-in loop: once the batch signature repeats, model(data) replays the staged hipGraphs of the whole step
            # (cubercnn/solver/autoreplay.py); callers that pre-stage their batch (bench.py, GraphedPipelined) pass `packed`.
            # (Asked BEFORE the filter-transform scope opens: a replayed step carries its own transforms, the scope's eager launch --
            # 260 MB of traffic, 0.25 ms of host time -- would be thrown away)
            out = auto.forward(batched_inputs)
            if out is not None:
                return out
        if not self.training and packed is None:
            out = self._replayed_inference(batched_inputs, True)       # (likewise: the captured pass carries its own transforms)
            if out is not None:
                return out
        with HF.wino_weight_scope(self):   # every Winograd filter of the pass is transformed by one launch at its start
            return self._forward(batched_inputs, packed)

    def _forward(self, batched_inputs, packed=None):
        if not self.training:
            if not _EVAL_F22:
                return self.inference(batched_inputs, packed=packed, _replay_tried=True)
            # Inference runs every Winograd layer on the 16-point F(2x2,3x3) transform: the cube head's Gram-Schmidt amplifies feature
            # noise up to ~300x for near-parallel 6D pose vectors (tools/probes/pose_diag.py), and the 36-point transform of the
            # bottom-up is what that noise is made of -- worst pose / corner error of the full-size fixture against float64 2.9e-4
            # with F(4x4,3x3), 3.5e-5 without (north_star: 1e-4), for 4 % of the inference time.  Training keeps F(4x4,3x3): its
            # losses sit at 1e-7 and its gradients are judged against the fp32 reference's own distance to float64.
            from ....kernels import wino as _wino
            with _wino.f22_only():
                return self.inference(batched_inputs, packed=packed, _replay_tried=True)
        auto = self.__dict__.get("_omni_auto")         # (forward() has already asked it for a replay)
        packed_given = packed
        if packed is None:
            packed = self.prepack(batched_inputs)
        images = self.preprocess_image(batched_inputs, slot_hw=packed.image_hw if getattr(packed, "slotted", False) else None,
                                       hw_dev=getattr(packed, "image_hw", None))
        features = self.backbone(images.tensor)
        if getattr(self, "feature_cut", None) is not None:   # data-parallel two-phase backward (solver/graphed.py)
            features = self.feature_cut(features)
        self._bump_bn_counters()
        # the reference's call contracts (rcnn3d.py:50-70): list[Instances] ground truth, per-image intrinsics and scale
        # ratios.  This package's own components additionally take the packed device copy (`accepts_packed`), so the
        # ground truth crosses PCIe once; a reference-style component registered under the same registry gets the lists.
        gt_instances = [x["instances"].to(self.device) for x in batched_inputs] if not self._all_packed() else None
        Ks = [torch.FloatTensor(x["K"]) for x in batched_inputs]
        im_scales_ratio = [x["height"] / im[0] for x, im in zip(batched_inputs, images.image_sizes)]
        pk = {"targets": packed} if getattr(self.proposal_generator, "accepts_packed", False) else {}
        proposals, proposal_losses = self.proposal_generator(images, features, gt_instances, **pk)
        pk = {"packed": packed} if getattr(self.roi_heads, "accepts_packed", False) else {}
        _, detector_losses = self.roi_heads(images, features, proposals, Ks, im_scales_ratio, gt_instances, **pk)
        losses = HF.LossDict()
        losses.update(detector_losses)
        losses.update(proposal_losses)
        if has_event_storage():
            self.flush_logs(get_event_storage())
        if packed_given is None and auto is not None and not auto.busy:
            losses = auto.wrap_eager(losses)            # the loop holds one releasable node, not the iteration's graph
        if packed_given is None and getattr(self, "_omni_owns_exchange", False):
            from ...solver.ddp import tie_to_anchor     # loop-level call under the script's DDP wrapper (cubercnn/solver/ddp.py)
            tie_to_anchor(self, losses)
        return losses

    def _all_packed(self):
        return getattr(self.proposal_generator, "accepts_packed", False) and getattr(self.roi_heads, "accepts_packed", False)

    def _bump_bn_counters(self):
        """`num_batches_tracked += 1` of every training-mode BatchNorm in one multi-tensor launch (39 adds otherwise)."""
        from ..layers import BatchNorm2d
        bns = self.__dict__.get("_bn_modules")
        if bns is None:
            bns = self.__dict__["_bn_modules"] = [m for m in self.modules() if isinstance(m, BatchNorm2d)]
            for m in bns:
                m.defer_counter = True
        live = [m.num_batches_tracked for m in bns if m.training and m.track_running_stats and m.num_batches_tracked is not None]
        if live:
            from ....kernels import glue
            glue.bump_counters(live, 1)

    def flush_logs(self, storage):
        """One device->host readback for all logged scalars (the reference does ~16 .item() syncs)."""
        for m in (self.proposal_generator, self.roi_heads):
            if hasattr(m, "flush_logs"):       # (a reference-style component logs its scalars itself)
                m.flush_logs(storage)

    def _inference_device(self, batched_inputs, packed):
        """the device half of inference() for this package's own components: fixed shapes, no host synchronisation"""
        from ..roi_heads.inference import roi_heads_inference_device
        images = self.preprocess_image(batched_inputs, slot_hw=packed.image_hw if getattr(packed, "slotted", False) else None)
        features = self.backbone(images.tensor)
        proposals, _ = self.proposal_generator(images, features, None, targets=packed)
        return roi_heads_inference_device(self.roi_heads, [features[f] for f in self.roi_heads.in_features], proposals, packed)

    def _replayed_inference(self, batched_inputs, do_postprocess):
        """-> results of a pass replayed from its captured hipGraph, or None (meta_arch/infer_replay.py)"""
        if not (self._all_packed() and getattr(self.roi_heads, "replayable_inference", False)):
            return None
        rep = self.__dict__.get("_omni_infer")
        if rep is None:
            from .infer_replay import InferReplay
            rep = self.__dict__["_omni_infer"] = InferReplay(self)
        import contextlib
        from ....kernels import wino as _wino

        def context():
            st = contextlib.ExitStack()
            st.enter_context(HF.wino_weight_scope(self))
            if _EVAL_F22:
                st.enter_context(_wino.f22_only())
            return st
        got = rep.run(batched_inputs, context)
        if got is None:
            return None
        from ..roi_heads.inference import collect_detections, postprocess
        raw, sizes = got
        results = collect_detections(raw, sizes)
        return postprocess(results, batched_inputs, sizes) if do_postprocess else results

    def inference(self, batched_inputs, detected_instances=None, do_postprocess=True, packed=None, _replay_tried=False):
        assert not self.training
        from ..roi_heads.inference import postprocess
        if packed is None and detected_instances is None and not _replay_tried:
            out = self._replayed_inference(batched_inputs, do_postprocess)
            if out is not None:
                return out
        images = self.preprocess_image(batched_inputs)
        if packed is None:
            sizes = images.image_sizes
            packed = pack_targets(batched_inputs, sizes, getattr(self.roi_heads, "virtual_focal", 512.0), with_gt=False).to(self.device)
        features = self.backbone(images.tensor)
        Ks = [torch.FloatTensor(x["K"]) for x in batched_inputs]
        im_scales_ratio = [x["height"] / im[0] for x, im in zip(batched_inputs, images.image_sizes)]
        if any("oracle2D" in b for b in batched_inputs):       # rcnn3d.py:98-101: oracle 2D boxes go straight to the ROI heads
            proposals = [b["oracle2D"] for b in batched_inputs]
        else:
            pk = {"targets": packed} if getattr(self.proposal_generator, "accepts_packed", False) else {}
            proposals, _ = self.proposal_generator(images, features, None, **pk)
        pk = {"packed": packed} if getattr(self.roi_heads, "accepts_packed", False) else {}
        results, _ = self.roi_heads(images, features, proposals, Ks, im_scales_ratio, None, **pk)
        if do_postprocess:
            return postprocess(results, batched_inputs, images.image_sizes)
        return results


def build_model(cfg, priors=None):
    meta_arch = cfg.MODEL.META_ARCHITECTURE
    model = META_ARCH_REGISTRY.get(meta_arch)(cfg, priors=priors)
    model.to(torch.device(cfg.MODEL.DEVICE))
    import os
    from ...solver import ddp
    if ddp.world() > 1 and os.environ.get("OMNI_OWN_EXCHANGE", "1") != "0":
        # the script will wrap this model in torch's DistributedDataParallel (tools/train_net.py:449-454): keep the gradient
        # exchange with the optimizer's flat bucket instead of DDP's per-tensor reducer (cubercnn/solver/ddp.py)
        ddp.prepare_for_ddp(model)
    return model


def build_backbone(cfg, input_shape=None, priors=None):
    if input_shape is None:
        input_shape = ShapeSpec(channels=len(cfg.MODEL.PIXEL_MEAN))
    backbone = BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, input_shape, priors)
    from ..backbone.fpn import Backbone
    assert isinstance(backbone, Backbone)
    return backbone
