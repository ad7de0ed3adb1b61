"""Inference half of the ROI heads: `FastRCNNOutputs.inference` / `fast_rcnn_inference_single_image`
(/root/reference/cubercnn/modeling/roi_heads/fast_rcnn.py:57-143), the eval branch of
`ROIHeads3D._forward_cube` (roi_heads.py:353-357, 771-824) and detectron2's `_postprocess`.

The heavy parts run in the HIP kernels (ROIAlign, the FC GEMMs, NMS, the fused cube decode); the
per-image candidate filtering (`scores > thresh` -> nonzero) is data-dependent compaction done with
torch indexing on a few thousand elements.  Batched inference fusion is SURVEY.md 8(f) item 3 ("next")."""
import math

import torch

from ....d2.structures import Boxes, Instances
from ....kernels import det, select

_SCALE_CLAMP = math.log(1000.0 / 16)


def _apply_deltas(deltas, boxes, weights):
    """detectron2 Box2BoxTransform.apply_deltas for all classes: deltas (R, 4K), boxes (R, 4)."""
    wx, wy, ww, wh = weights
    widths, heights = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    ctr_x, ctr_y = boxes[:, 0] + 0.5 * widths, boxes[:, 1] + 0.5 * heights
    dx, dy = deltas[:, 0::4] / wx, deltas[:, 1::4] / wy
    dw, dh = (deltas[:, 2::4] / ww).clamp(max=_SCALE_CLAMP), (deltas[:, 3::4] / wh).clamp(max=_SCALE_CLAMP)
    pcx, pcy = dx * widths[:, None] + ctr_x[:, None], dy * heights[:, None] + ctr_y[:, None]
    pw, ph = torch.exp(dw) * widths[:, None], torch.exp(dh) * heights[:, None]
    return torch.stack((pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph), dim=-1)   # (R, K, 4)


@torch.no_grad()
def roi_heads_inference(heads, images, feats, proposals, packed):
    K = heads.num_classes
    pred_boxes_all, count = proposals.boxes, proposals.count
    B, P = pred_boxes_all.shape[:2]
    rois = pred_boxes_all.reshape(B * P, 4)
    bidx = heads._batch_index(B, P, rois.device)
    x = heads.box_pooler(feats, rois, bidx)
    pred = heads.box_predictor(heads.box_head(x))
    probs = torch.softmax(pred[:, : K + 1], dim=-1).reshape(B, P, K + 1)
    boxes = _apply_deltas(pred[:, K + 1: K + 1 + 4 * K], rois, heads.box_predictor.box2box_weights).reshape(B, P, K, 4)
    counts = count.tolist()
    thr, nms_thr, topk = heads.box_predictor.test_score_thresh, heads.box_predictor.test_nms_thresh, heads.box_predictor.test_topk_per_image
    per_image = []
    for n in range(B):
        h, w = images.image_sizes[n]
        bx, sc = boxes[n, : counts[n]], probs[n, : counts[n]]
        valid = torch.isfinite(bx).all(dim=2).all(dim=1) & torch.isfinite(sc).all(dim=1)
        bx, sc = bx[valid], sc[valid][:, :-1]
        bx = torch.stack((bx[..., 0].clamp(0, w), bx[..., 1].clamp(0, h), bx[..., 2].clamp(0, w), bx[..., 3].clamp(0, h)), dim=-1)
        mask = sc > thr
        inds = mask.nonzero()
        cb, cs, full = bx[mask], sc[mask], sc[inds[:, 0]]
        if cs.numel() > 8192:   # NMS kernel capacity; keep the best candidates
            top = torch.topk(cs, 8192)[1]
            cb, cs, inds, full = cb[top], cs[top], inds[top], full[top]
        order = torch.sort(cs, descending=True, stable=True)[1]
        cb, cs, inds, full = cb[order], cs[order], inds[order], full[order]
        if cs.numel() > 0:
            # per-class NMS as one problem: class-wise coordinate offset (torchvision batched_nms)
            off = inds[:, 1].to(cb.dtype) * (cb.max() + 1)
            keep = select.nms_sorted((cb + off[:, None])[None].contiguous(), nms_thr)[0].bool()
            keep = keep.nonzero().squeeze(1)[:topk]
        else:
            keep = torch.zeros(0, dtype=torch.long, device=cb.device)
        inst = Instances((h, w))
        inst.pred_boxes = Boxes(cb[keep])
        inst.scores = cs[keep]
        inst.scores_full = full[keep]
        inst.pred_classes = inds[keep, 1]
        per_image.append(inst)
    # ---- cube head on the kept detections (roi_heads.py:353-357, 771-819)
    nper = [len(i) for i in per_image]
    if sum(nper) > 0:
        dboxes = torch.cat([i.pred_boxes.tensor for i in per_image]).contiguous()
        dcls = torch.cat([i.pred_classes for i in per_image]).to(torch.int32).contiguous()
        dimg = torch.cat([torch.full((k,), n, dtype=torch.int32, device=dboxes.device) for n, k in enumerate(nper)]).contiguous()
        xc = heads.cube_pooler(feats, dboxes, dimg)
        head = heads.cube_head(xc)
        priors = heads.priors_dims_per_cat.detach().reshape(K, 2, 3).contiguous()
        cube3d, pose, verts = det.cube_decode(head.contiguous(), K, dboxes, dcls, dimg, packed.Ks, packed.v2r, packed.ratio, priors)
        o = 0
        for inst, k in zip(per_image, nper):
            c3 = cube3d[o:o + k]
            inst.scores = (inst.scores * c3[:, 8]) ** 0.5                 # roi_heads.py:800-801
            inst.pred_bbox3D = verts[o:o + k]
            inst.pred_center_cam = c3[:, :3]
            inst.pred_center_2D = c3[:, 6:8]
            inst.pred_dimensions = c3[:, 3:6]
            inst.pred_pose = pose[o:o + k]
            o += k
    return per_image


def postprocess(instances, batched_inputs, image_sizes):
    """detectron2 GeneralizedRCNN._postprocess / detector_postprocess: rescale 2D boxes to the original
    resolution, clip, drop empty boxes; 3D fields pass through."""
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
