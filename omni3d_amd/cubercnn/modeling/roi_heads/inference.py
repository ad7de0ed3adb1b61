"""Inference half of the ROI heads: `FastRCNNOutputs.inference` / `fast_rcnn_inference_single_image`
(/root/reference/cubercnn/modeling/roi_heads/fast_rcnn.py:57-143), the eval branch of
`ROIHeads3D._forward_cube` (roi_heads.py:353-357, 771-824) and detectron2's `_postprocess`.

Everything runs in HIP kernels on fixed shapes for the whole batch (ROIAlign, the FC GEMMs, csrc/infer.hip, the top-k and
NMS kernels, the fused cube decode); the host reads back only the per-image detection counts."""
import torch

from ....d2.structures import Boxes, Instances
from ....kernels import det, select

@torch.no_grad()
def roi_heads_inference(heads, images, feats, proposals, packed):
    """Whole batch, fixed shapes, ONE device->host copy (the per-image detection counts) at the very end: score / decode /
    clip / threshold (csrc/infer.hip), stable candidate sort (top-k kernel), per-class NMS as one problem per image, the
    DETECTIONS_PER_IMAGE best into fixed slots, then the cube head + fused decode on those slots."""
    return collect_detections(roi_heads_inference_device(heads, feats, proposals, packed), images.image_sizes)


@torch.no_grad()
def roi_heads_inference_device(heads, feats, proposals, packed):
    """the device half: fixed-shape tensors for the whole batch, no host synchronisation (capturable: meta_arch/infer_replay.py)"""
    K = heads.num_classes
    pred_boxes_all, count = proposals.boxes, proposals.count
    B, P = pred_boxes_all.shape[:2]
    rois = pred_boxes_all.reshape(B * P, 4).contiguous()
    bidx = heads._batch_index(B, P, rois.device)
    x = heads.box_pooler(feats, rois, bidx)
    pred = heads.box_predictor(heads.box_head(x)).contiguous()
    bp = heads.box_predictor
    topk = bp.test_topk_per_image
    cap = min(det.DET_MAX_CANDIDATES, P * K)
    scores, probs, boxes = det.det_scores(pred, rois, count, packed.image_hw, B, P, K, bp.box2box_weights, bp.test_score_thresh)
    vals, idx = select.topk_rows(scores, cap)                      # stable descending order of the row-major (roi, class) list
    nms_boxes, valid = det.det_nms_boxes(boxes, vals, idx, B, P * K, K, cap)
    keep = select.nms_sorted(nms_boxes, bp.test_nms_thresh, None, valid)
    dbox, dscore, dcls, droi, dcount = det.det_compact(keep, valid, vals, idx, boxes, B, P * K, K, cap, topk)
    # ---- cube head on the fixed (B, topk) slots (roi_heads.py:353-357, 771-819); unused slots hold a dummy box
    dboxes, dcl = dbox.view(B * topk, 4), dcls.view(-1)
    dimg = heads._batch_index(B, topk, dboxes.device)
    xc = heads.cube_pooler(feats, heads.scale_proposals(dboxes), dimg)
    head = heads.cube_head(xc)
    priors = heads.priors_dims_per_cat.detach().reshape(K, 2, 3).contiguous()
    cube3d, pose, verts = det.cube_decode(head.contiguous(), K, dboxes, dcl, dimg, packed.Ks, packed.v2r, packed.ratio, priors,
                                           heads.cube_mode, heads.clusters())
    final = (dscore.view(-1) * cube3d[:, 8]) ** 0.5                                               # roi_heads.py:800-801
    full = torch.gather(probs.view(B, P, K), 1, droi.long()[:, :, None].expand(-1, -1, K))        # scores of all classes per kept roi
    return {"dbox": dbox, "final": final, "full": full, "dcls": dcls, "verts": verts, "cube3d": cube3d, "pose": pose, "dcount": dcount}


def collect_detections(raw, image_sizes):
    """the host half: ONE synchronisation (the per-image detection counts), then list[Instances] of row slices of the batch tensors"""
    dbox, final, full, dcls, verts, cube3d, pose = (raw[k] for k in ("dbox", "final", "full", "dcls", "verts", "cube3d", "pose"))
    B, topk = dbox.shape[:2]
    counts = raw["dcount"].tolist()                                                               # the one host sync
    per_image = []
    for n in range(B):
        k, o = counts[n], n * topk
        inst = Instances(tuple(image_sizes[n]))
        inst.pred_boxes = Boxes(dbox[n, :k])
        inst.scores = final[o:o + k]
        inst.scores_full = full[n, :k]
        inst.pred_classes = dcls[n, :k].long()
        inst.pred_bbox3D = verts[o:o + k]
        inst.pred_center_cam = cube3d[o:o + k, :3]
        inst.pred_center_2D = cube3d[o:o + k, 6:8]
        inst.pred_dimensions = cube3d[o:o + k, 3:6]
        inst.pred_pose = pose[o:o + k]
        inst._omni_slots = (dbox, n, k)           # postprocess(): all images' boxes rescaled / clipped / tested in one pass
        per_image.append(inst)
    return per_image


@torch.no_grad()
def roi_heads_oracle2d(heads, images, feats, oracles, packed):
    """The eval branch of ROIHeads3D.forward for given 2D boxes (roi_heads.py:228-240, then _forward_cube :353-357, 771-819):
    `oracles` = per image {'gt_bbox2D': (n, 4) boxes at network resolution, 'gt_classes': (n,)}; the detections are those boxes with
    score sqrt(1 * confidence) and the cube head's 3D outputs.  One fixed-shape pass for the whole batch."""
    K = heads.num_classes
    dev = images.tensor.device
    B = len(oracles)
    counts = [int(len(o["gt_classes"])) for o in oracles]
    P = max(counts + [1])
    boxes = torch.zeros((B, P, 4), dtype=torch.float32, device=dev)
    boxes[:, :, 2:] = 1.0                                                # unused slots hold a dummy box
    cls = torch.zeros((B, P), dtype=torch.int32, device=dev)
    for n, o in enumerate(oracles):
        if counts[n]:
            boxes[n, :counts[n]] = torch.as_tensor(o["gt_bbox2D"], dtype=torch.float32).reshape(-1, 4).to(dev)
            cls[n, :counts[n]] = torch.as_tensor(o["gt_classes"]).to(dev).int()
    dboxes, dcl = boxes.view(B * P, 4), cls.view(-1)
    dimg = heads._batch_index(B, P, dev)
    head = heads.cube_head(heads.cube_pooler(feats, heads.scale_proposals(dboxes), dimg))
    priors = heads.priors_dims_per_cat.detach().reshape(K, 2, 3).contiguous()
    cube3d, pose, verts = det.cube_decode(head.contiguous(), K, dboxes, dcl, dimg, packed.Ks, packed.v2r, packed.ratio, priors,
                                           heads.cube_mode, heads.clusters())
    final = cube3d[:, 8] ** 0.5                                          # (ones * cube_3D[:, -1]) ** (1 / 2), roi_heads.py:239, 800-801
    per_image = []
    for n, o in enumerate(oracles):
        k, s = counts[n], n * P
        inst = Instances(tuple(images.image_sizes[n]))
        inst.pred_boxes = Boxes(boxes[n, :k])
        inst.pred_classes = torch.as_tensor(o["gt_classes"]).to(dev)
        inst.scores = final[s:s + k]
        inst.pred_bbox3D = verts[s:s + k]
        inst.pred_center_cam = cube3d[s:s + k, :3]
        inst.pred_center_2D = cube3d[s:s + k, 6:8]
        inst.pred_dimensions = cube3d[s:s + k, 3:6]
        inst.pred_pose = pose[s:s + k]
        per_image.append(inst)
    return per_image


def postprocess(instances, batched_inputs, image_sizes):
    """detectron2 GeneralizedRCNN._postprocess / detector_postprocess: rescale 2D boxes to the original
    resolution, clip, drop empty boxes; 3D fields pass through."""
    slots = [getattr(res, "_omni_slots", None) for res in instances]
    if slots and all(s is not None and s[0] is slots[0][0] and s[1] == n for n, s in enumerate(slots)):
        return _postprocess_slots(instances, batched_inputs, image_sizes, slots[0][0])
    out = []
    for res, info, size in zip(instances, batched_inputs, image_sizes):
        H, W = info.get("height", size[0]), info.get("width", size[1])
        sx, sy = W / res.image_size[1], H / res.image_size[0]
        r = Instances((H, W), **res.get_fields())
        b = r.pred_boxes.tensor.clone()
        b[:, 0::2] *= sx
        b[:, 1::2] *= sy
        b = torch.stack((b[:, 0].clamp(0, W), b[:, 1].clamp(0, H), b[:, 2].clamp(0, W), b[:, 3].clamp(0, H)), dim=1)
        r.pred_boxes = Boxes(b)
        out.append({"instances": r[r.pred_boxes.nonempty()]})
    return out


@torch.no_grad()
def _postprocess_slots(instances, batched_inputs, image_sizes, dbox):
    """postprocess() for the results of roi_heads_inference: their boxes are row slices of ONE (B, topk, 4) tensor, so the rescale,
    the clip and the empty-box test run once for the batch (the same multiplications and clamps, in the same order), the mask comes
    back in one copy, and an image whose boxes are all non-empty -- every image, normally -- keeps its fields as the views they are.
    Per image the reference's form costs ~25 launches (boolean indexing of nine fields = nine nonzero + gather pairs)."""
    HW = [(info.get("height", size[0]), info.get("width", size[1])) for info, size in zip(batched_inputs, image_sizes)]
    sc = torch.tensor([[W / res.image_size[1], H / res.image_size[0]] * 2 for (H, W), res in zip(HW, instances)], dtype=torch.float32)
    lim = torch.tensor([[W, H, W, H] for H, W in HW], dtype=torch.float32)
    both = torch.stack([sc, lim]).to(dbox.device, non_blocking=True)               # one copy
    b = torch.minimum((dbox * both[0][:, None, :]).clamp_(min=0), both[1][:, None, :])
    ok = ((b[..., 2] - b[..., 0] > 0) & (b[..., 3] - b[..., 1] > 0)).cpu()
    out = []
    for n, (res, (H, W)) in enumerate(zip(instances, HW)):
        k = res._omni_slots[2]
        r = Instances((H, W), **res.get_fields())
        r.pred_boxes = Boxes(b[n, :k])
        keep = ok[n, :k]
        out.append({"instances": r if bool(keep.all()) else r[keep.to(dbox.device)]})
    return out
