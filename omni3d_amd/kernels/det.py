"""Launchers for the detection-logic kernels: csrc/rpn_roi.hip, roi_align.hip, box_loss.hip,
cube_head.hip, optim.hip.  Thin: allocate outputs, pass raw pointers + the current stream."""
import ctypes
import math

import torch

from .. import lib as _lib

_P = ctypes.c_void_p


def _ptrs(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


def _ints(vals):
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


def _floats(vals):
    return (ctypes.c_float * len(vals))(*[float(v) for v in vals])


def _cast(a):
    return ctypes.cast(a, _P)


def _dev(*ts):
    return _lib.check_device(*ts)


def _empty(shape, dtype, like):
    return torch.empty(shape, dtype=dtype, device=like.device)


def pairwise_iou(boxes1, boxes2, mode="iou"):
    boxes1, boxes2 = boxes1.contiguous().float(), boxes2.contiguous().float()
    L = _dev(boxes1, boxes2)
    N, M = boxes1.shape[0], boxes2.shape[0]
    out = _empty((N, M), torch.float32, boxes1)
    L.call("omni_pairwise_iou", _lib.ptr(boxes1), N, _lib.ptr(boxes2), M, 0 if mode == "iou" else 1, _lib.ptr(out),
           _lib.stream_of(boxes1))
    return out


def rpn_match(anchors, gt, gt_off, expo, thresholds=(0.05, 0.05), labels=(0, -1, 1), allow_low_quality=True, eps=1e-4, draw=None, B=None):
    """-> dict(matched_val, matched_idx, match_label, gt_best_idx, key_pos, key_neg).
    expo (B, A): Exp(1) variates of the sampling keys, or None with draw = a glue.DrawState and B = batch size: the kernel draws them
    itself (csrc/philox.h) -- no array, no generator launches."""
    L = _dev(anchors, gt, gt_off, expo)
    A, G = anchors.shape[0], gt.shape[0]
    B = expo.shape[0] if expo is not None else int(B)
    o = {
        "matched_val": _empty((B, A), torch.float32, anchors), "matched_idx": _empty((B, A), torch.int32, anchors),
        "match_label": _empty((B, A), torch.int8, anchors), "gt_best_idx": _empty((max(G, 1),), torch.int32, anchors),
        "keys": _empty((2 * B, A), torch.float32, anchors),   # rows [0, B) positive keys, [B, 2B) negative keys
    }
    bits = _empty((max(G, 1),), torch.int32, anchors)
    state, ticket = draw.tensors(anchors.device) if expo is None else (None, None)
    L.call("omni_rpn_match_draw", _lib.ptr(anchors), A, _lib.ptr(gt), _lib.ptr(gt_off), B, G, float(thresholds[0]),
           float(thresholds[1]), int(labels[0]), int(labels[1]), int(labels[2]), int(allow_low_quality), _lib.ptr(expo),
           _lib.ptr(state), _lib.ptr(ticket), float(eps), _lib.ptr(o["matched_val"]), _lib.ptr(o["matched_idx"]), _lib.ptr(o["match_label"]),
           _lib.ptr(bits), _lib.ptr(o["gt_best_idx"]), _lib.ptr(o["keys"][:B]), _lib.ptr(o["keys"][B:]), _lib.stream_of(anchors))
    o["key_pos"], o["key_neg"] = o["keys"][:B], o["keys"][B:]
    return o


def rpn_finalize_labels(anchors, gt_off, ign, ign_off, match, pos_val, pos_idx, neg_val, neg_idx, batch_per_image,
                        ignore_thresh):
    L = _dev(anchors, gt_off, ign, ign_off, pos_val, pos_idx, neg_val, neg_idx)
    A, B = anchors.shape[0], pos_val.shape[0]
    labels = _empty((B, A), torch.int8, anchors)
    counts = _empty((B, 2), torch.int32, anchors)
    L.call("omni_rpn_finalize_labels", _lib.ptr(anchors), A, B, _lib.ptr(gt_off), _lib.ptr(ign), _lib.ptr(ign_off),
           _lib.ptr(match["match_label"]), _lib.ptr(match["gt_best_idx"]), _lib.ptr(pos_val), _lib.ptr(pos_idx),
           _lib.ptr(neg_val), _lib.ptr(neg_idx), pos_val.shape[1], neg_val.shape[1], int(batch_per_image),
           float(ignore_thresh), _lib.ptr(labels), _lib.ptr(counts), _lib.stream_of(anchors))
    return labels, counts


class LevelPack:
    """Host-side description of the per-level RPN head tensors (B, H, W, 16) NHWC."""

    def __init__(self, tensors_nhwc):
        self.tensors = [t for t in tensors_nhwc]
        for t in self.tensors:
            assert t.is_contiguous() and t.shape[-1] == 16
        self.hw = [t.shape[1] * t.shape[2] for t in self.tensors]
        self.B = self.tensors[0].shape[0]
        self.A = 3 * sum(self.hw)

    def args(self):
        p, h = _ptrs(self.tensors), _ints(self.hw)
        self._keep = (p, h)
        return _cast(p), _cast(h), len(self.tensors)


def _longs(vals):
    return (ctypes.c_longlong * len(vals))(*[int(v) for v in vals])


HEAD16_C = 256          # input width the fused RPN head kernels (csrc/rpn_head.hip) are written for
_HEAD16_ROWS = 512


def head16_fwd(ts, w_obj, b_obj, w_del, b_del):
    """ts: per-level (B, H, W, 256) NHWC activations; w_obj (3, 256), w_del (12, 256) row-major -> per-level (B, H, W, 16)
    [3 logits | 12 deltas | 0] of detectron2's StandardRPNHead, all levels in one launch"""
    L = _dev(*ts, w_obj, b_obj, w_del, b_del)
    ys = [_empty(t.shape[:3] + (16,), torch.float32, t) for t in ts]
    tp, yp, px = _ptrs(ts), _ptrs(ys), _longs([t.shape[0] * t.shape[1] * t.shape[2] for t in ts])
    L.call("omni_rpn_head16_fwd", _cast(tp), _cast(px), len(ts), _lib.ptr(w_obj), _lib.ptr(b_obj), _lib.ptr(w_del), _lib.ptr(b_del), _cast(yp),
           _lib.stream_of(ts[0]))
    return ys


def head16_dgrad(dys, ts, w_obj, w_del, relu_mask=True):
    """-> per-level gradients wrt ts, zeroed where ts <= 0 (relu_mask: ts are ReLU outputs and the gradient wrt the ReLU INPUT is wanted)"""
    L = _dev(*dys, *ts, w_obj, w_del)
    dts = [torch.empty_like(t) for t in ts]
    dp, tp, op, px = _ptrs(dys), _ptrs(ts), _ptrs(dts), _longs([t.shape[0] * t.shape[1] * t.shape[2] for t in ts])
    L.call("omni_rpn_head16_dgrad", _cast(dp), _cast(tp), _cast(px), len(ts), _lib.ptr(w_obj), _lib.ptr(w_del), int(bool(relu_mask)), _cast(op),
           _lib.stream_of(ts[0]))
    return dts


def head16_wgrad(dys, ts, accum_into=None):
    """-> (dw_obj (3, 256), db_obj (3), dw_del (12, 256), db_del (12)); accum_into: the same four as contiguous buffers (None entries
    allowed) that the sums are ADDED to instead (returns None for those)"""
    L = _dev(*dys, *ts)
    ref = ts[0]
    partial = _empty((_HEAD16_ROWS * (15 * HEAD16_C + 16),), torch.float32, ref)
    fresh = [_empty(s, torch.float32, ref) for s in ((3, HEAD16_C), (3,), (12, HEAD16_C), (12,))]
    dp, tp, px = _ptrs(dys), _ptrs(ts), _longs([t.shape[0] * t.shape[1] * t.shape[2] for t in ts])
    if accum_into is not None and all(a is not None for a in accum_into):
        for a in accum_into:
            assert a.is_contiguous()
        L.call("omni_rpn_head16_wgrad", _cast(dp), _cast(tp), _cast(px), len(ts), _lib.ptr(partial), _HEAD16_ROWS, *[_lib.ptr(a) for a in accum_into],
               1, _lib.stream_of(ref))
        return (None, None, None, None)
    L.call("omni_rpn_head16_wgrad", _cast(dp), _cast(tp), _cast(px), len(ts), _lib.ptr(partial), _HEAD16_ROWS, *[_lib.ptr(a) for a in fresh], 0,
           _lib.stream_of(ref))
    return tuple(fresh)


def rpn_gather_logits(pack):
    L = _dev(*pack.tensors)
    out = _empty((pack.B, pack.A), torch.float32, pack.tensors[0])
    L.call("omni_rpn_gather_logits", *pack.args(), pack.B, _lib.ptr(out), _lib.stream_of(out))
    return out


def rpn_loss_fwd(pack, anchors, labels, matched_idx, gt, gt_off, plain=False):
    """plain: MODEL.RPN.OBJECTNESS_UNCERTAINTY 'none' (0 / 1 objectness targets, unweighted L1) instead of 'IoUness'"""
    L = _dev(anchors, labels, matched_idx, gt, gt_off)
    sums = _empty((6,), torch.float64, anchors)
    L.call("omni_rpn_loss_plain_fwd" if plain else "omni_rpn_loss_fwd", *pack.args(), pack.B, _lib.ptr(anchors), _lib.ptr(labels), _lib.ptr(matched_idx),
           _lib.ptr(gt), _lib.ptr(gt_off), _lib.ptr(sums), _lib.stream_of(anchors))
    return sums


def rpn_loss_bwd(pack, anchors, labels, matched_idx, gt, gt_off, g_cls, g_loc, inv_norm, plain=False):
    L = _dev(anchors, labels, matched_idx, gt, gt_off, g_cls, g_loc)
    grads = [torch.empty_like(t) for t in pack.tensors]
    dp = _ptrs(grads)
    a = pack.args()
    L.call("omni_rpn_loss_plain_bwd" if plain else "omni_rpn_loss_bwd", a[0], _cast(dp), a[1], a[2], pack.B, _lib.ptr(anchors), _lib.ptr(labels),
           _lib.ptr(matched_idx), _lib.ptr(gt), _lib.ptr(gt_off), _lib.ptr(g_cls), _lib.ptr(g_loc), float(inv_norm),
           _lib.stream_of(anchors))
    return grads


def rpn_decode(pack, slot_level, idx, anchors, image_hw, min_size=0.0):
    L = _dev(slot_level, idx, anchors, image_hw)
    B, Ktot = idx.shape
    boxes = _empty((B, Ktot, 4), torch.float32, anchors)
    valid = _empty((B, Ktot), torch.int32, anchors)
    L.call("omni_rpn_decode", *pack.args(), B, Ktot, _lib.ptr(slot_level), _lib.ptr(idx), _lib.ptr(anchors),
           _lib.ptr(image_hw), float(math.log(1000.0 / 16)), float(min_size), _lib.ptr(boxes), _lib.ptr(valid),
           _lib.stream_of(anchors))
    return boxes, valid


def rpn_mask_scores(scores, keep):
    """-> scores where keep != 0 else -inf (same shape)"""
    L = _dev(scores, keep)
    assert scores.is_contiguous() and keep.is_contiguous() and keep.dtype == torch.int32 and scores.shape == keep.shape
    out = _empty(tuple(scores.shape), torch.float32, scores)
    L.call("omni_rpn_mask_scores", _lib.ptr(scores), _lib.ptr(keep), scores.numel(), _lib.ptr(out), _lib.stream_of(scores))
    return out


def rpn_collect(boxes, top_v, top_i):
    """boxes (B, N, 4), top_v / top_i (B, P) -> (prop (B, P, 4), count (B) int32)"""
    L = _dev(boxes, top_v, top_i)
    B, N, _ = boxes.shape
    P = top_v.shape[1]
    assert boxes.is_contiguous() and top_v.is_contiguous() and top_i.is_contiguous() and top_i.dtype == torch.int32
    prop = _empty((B, P, 4), torch.float32, boxes)
    count = _empty((B,), torch.int32, boxes)
    L.call("omni_rpn_collect", _lib.ptr(boxes), _lib.ptr(top_v), _lib.ptr(top_i), B, N, P, _lib.ptr(prop), _lib.ptr(count), _lib.stream_of(boxes))
    return prop, count


ROI_MAXC = 2048


def roi_sample(prop_boxes, prop_count, gt, gt_cls, gt_off, ign, ign_off, expo, iou_thr, ignore_thresh, num_classes,
               batch_per_image, positive_fraction, append_gt=True, eps=1e-4, draw=None, first=0):
    """-> (boxes, cls, gt row, iou, counts) of the sampled ROIs, + (rows clamped to >= 0,) + the contiguous copies (boxes, cls, clamped
    rows) of the first `first` slots per image when first > 0.  expo (B, ROI_MAXC) Exp(1) variates, or None with draw = a
    glue.DrawState: drawn inside the kernel (csrc/philox.h)."""
    L = _dev(prop_boxes, prop_count, gt, gt_cls, gt_off, ign, ign_off, expo)
    B, pmax = prop_boxes.shape[0], prop_boxes.shape[1]
    assert expo is None or expo.shape == (B, ROI_MAXC)
    o_boxes = _empty((B, batch_per_image, 4), torch.float32, prop_boxes)
    o_cls = _empty((B, batch_per_image), torch.int32, prop_boxes)
    o_gt = _empty((B, batch_per_image), torch.int32, prop_boxes)
    o_iou = _empty((B, batch_per_image), torch.float32, prop_boxes)
    o_cnt = _empty((B, 2), torch.int32, prop_boxes)
    o_row = _empty((B, batch_per_image), torch.int32, prop_boxes)
    first = int(min(first, batch_per_image))
    f_boxes = _empty((B, first, 4), torch.float32, prop_boxes) if first > 0 else None
    f_cls = _empty((B, first), torch.int32, prop_boxes) if first > 0 else None
    f_row = _empty((B, first), torch.int32, prop_boxes) if first > 0 else None
    state, ticket = draw.tensors(prop_boxes.device) if expo is None else (None, None)
    L.call("omni_roi_sample_draw", _lib.ptr(prop_boxes), _lib.ptr(prop_count), B, pmax, _lib.ptr(gt), _lib.ptr(gt_cls),
           _lib.ptr(gt_off), _lib.ptr(ign), _lib.ptr(ign_off), _lib.ptr(expo), _lib.ptr(state), _lib.ptr(ticket), float(iou_thr),
           float(ignore_thresh), float(eps), int(num_classes), int(batch_per_image), int(batch_per_image * positive_fraction),
           int(append_gt), _lib.ptr(o_boxes), _lib.ptr(o_cls), _lib.ptr(o_gt), _lib.ptr(o_iou), _lib.ptr(o_cnt), _lib.ptr(o_row), first,
           _lib.ptr(f_boxes), _lib.ptr(f_cls), _lib.ptr(f_row), _lib.stream_of(prop_boxes))
    return o_boxes, o_cls, o_gt, o_iou, o_cnt, o_row, (f_boxes, f_cls, f_row)


def roi_levels(rois, min_level=2, max_level=6, canonical_size=224.0, canonical_level=4):
    L = _dev(rois)
    R = rois.shape[0]
    lv = _empty((R,), torch.int32, rois)
    L.call("omni_roi_levels", _lib.ptr(rois), R, min_level, max_level, float(canonical_size), canonical_level, _lib.ptr(lv),
           _lib.stream_of(rois))
    return lv


def _feat_args(feats_nhwc, scales):
    p = _ptrs(feats_nhwc)
    hw = _ints([v for t in feats_nhwc for v in (t.shape[1], t.shape[2])])
    sc = _floats(scales)
    return (p, hw, sc), (_cast(p), _cast(hw), _cast(sc), len(feats_nhwc))


def roi_align_fwd(feats_nhwc, scales, rois, batch_idx, levels, P):
    """feats: list of (B,H,W,C) contiguous; -> (R, P, P, C)."""
    L = _dev(rois, batch_idx, levels, *feats_nhwc)
    R, C = rois.shape[0], feats_nhwc[0].shape[3]
    out = _empty((R, P, P, C), torch.float32, rois)
    keep, a = _feat_args(feats_nhwc, scales)
    L.call("omni_roi_align_fwd", *a, _lib.ptr(rois), _lib.ptr(batch_idx), _lib.ptr(levels), R, P, C, _lib.ptr(out),
           _lib.stream_of(rois))
    return out


def roi_align_fwd_mode(feats_nhwc, scales, rois, batch_idx, levels, P, aligned):
    """roi_align_fwd with torchvision's `aligned` switch (False = detectron2 POOLER_TYPE "ROIAlign")"""
    L = _dev(rois, batch_idx, levels, *feats_nhwc)
    R, C = rois.shape[0], feats_nhwc[0].shape[3]
    out = _empty((R, P, P, C), torch.float32, rois)
    keep, a = _feat_args(feats_nhwc, scales)
    L.call("omni_roi_align_fwd_mode", *a, _lib.ptr(rois), _lib.ptr(batch_idx), _lib.ptr(levels), R, P, C, int(bool(aligned)), _lib.ptr(out),
           _lib.stream_of(rois))
    return out


def roi_align_bwd_mode(dfeats_nhwc, scales, rois, batch_idx, levels, P, aligned, dout):
    """atomic backward of roi_align_fwd_mode into ZEROED dfeats"""
    L = _dev(rois, batch_idx, levels, dout, *dfeats_nhwc)
    R, C = rois.shape[0], dfeats_nhwc[0].shape[3]
    keep, a = _feat_args(dfeats_nhwc, scales)
    L.call("omni_roi_align_bwd_mode", *a, _lib.ptr(rois), _lib.ptr(batch_idx), _lib.ptr(levels), R, P, C, int(bool(aligned)), _lib.ptr(dout),
           _lib.stream_of(rois))


def roi_pool_fwd(feats_nhwc, scales, rois, batch_idx, levels, P):
    """torchvision roi_pool level by level (detectron2 POOLER_TYPE "ROIPool") -> (out (R, P, P, C), argmax (R, P, P, C) int32)"""
    L = _dev(rois, batch_idx, levels, *feats_nhwc)
    R, C = rois.shape[0], feats_nhwc[0].shape[3]
    out = _empty((R, P, P, C), torch.float32, rois)
    arg = _empty((R, P, P, C), torch.int32, rois)
    keep, a = _feat_args(feats_nhwc, scales)
    L.call("omni_roi_pool_fwd", *a, _lib.ptr(rois), _lib.ptr(batch_idx), _lib.ptr(levels), R, P, C, _lib.ptr(out), _lib.ptr(arg),
           _lib.stream_of(rois))
    return out, arg


def roi_pool_bwd(dfeats_nhwc, scales, batch_idx, levels, P, dout, argmax):
    """dout lands on the argmax pixels: fp32 atomics into ZEROED dfeats"""
    L = _dev(batch_idx, levels, dout, argmax, *dfeats_nhwc)
    R, C = dout.shape[0], dfeats_nhwc[0].shape[3]
    keep, a = _feat_args(dfeats_nhwc, scales)
    L.call("omni_roi_pool_bwd", *a, _lib.ptr(batch_idx), _lib.ptr(levels), R, P, C, _lib.ptr(dout), _lib.ptr(argmax), _lib.stream_of(dout))


def roi_align_fwd2(feats_nhwc, scales, rois, batch_idx, levels, P, per_image, first):
    """-> (out (R, P, P, C), out2 ((R // per_image) * first, P, P, C) = the first `first` ROIs of every block of `per_image`), one pass"""
    L = _dev(rois, batch_idx, levels, *feats_nhwc)
    R, C = rois.shape[0], feats_nhwc[0].shape[3]
    out = _empty((R, P, P, C), torch.float32, rois)
    out2 = _empty(((R // per_image) * first, P, P, C), torch.float32, rois)
    keep, a = _feat_args(feats_nhwc, scales)
    L.call("omni_roi_align_fwd2", *a, _lib.ptr(rois), _lib.ptr(batch_idx), _lib.ptr(levels), R, P, C, _lib.ptr(out), _lib.ptr(out2),
           int(per_image), int(first), _lib.stream_of(rois))
    return out, out2


def roi_align_bwd_deterministic(P, R, C=256):
    """the owner-computes backward (omni_roi_align_bwd_det) serves the pooler resolution and FPN width of every Cube R-CNN config"""
    from . import detmode as _det
    return _det.on() and P == 7 and R <= 4096 and C <= 256


def roi_align_bwd_det(dfeats_nhwc, scales, rois, batch_idx, levels, P, dout, dout2=None, per_image=0, first=0):
    """OVERWRITES dfeats (any content, e.g. torch.empty): every element written once, ROI contributions added in ROI order"""
    L = _dev(rois, batch_idx, levels, dout, dout2, *dfeats_nhwc)
    R, C, B = rois.shape[0], dfeats_nhwc[0].shape[3], dfeats_nhwc[0].shape[0]
    keep, a = _feat_args(dfeats_nhwc, scales)
    assert (dout is None or dout.is_contiguous()) and (dout2 is None or dout2.is_contiguous())
    from . import detmode as _det
    args = (*a, B, _lib.ptr(rois), _lib.ptr(batch_idx), _lib.ptr(levels), R, P, C, _lib.ptr(dout), _lib.ptr(dout2), int(per_image), int(first))
    plan, addr = _det.new_plan()
    L.call("omni_roi_align_bwd_det", *args, None, 0, None, 0, addr, _lib.stream_of(rois))
    ws, wsf, ctr, nctr = _det.workspace(rois, plan)
    L.call("omni_roi_align_bwd_det", *args, _lib.ptr(ws), wsf, _lib.ptr(ctr), nctr, None, _lib.stream_of(rois))


def roi_align_bwd(dfeats_nhwc, scales, rois, batch_idx, levels, P, dout, dout2=None, per_image=0, first=0):
    """accumulates into dfeats (caller-zeroed or holding other gradients).  dout2 ((R / per_image) * first, P, P, C): a second gradient
    for the first `first` ROIs of every block of `per_image`, added on the fly (P == 7); dout may then be None."""
    L = _dev(rois, batch_idx, levels, dout, dout2, *dfeats_nhwc)
    R, C = rois.shape[0], dfeats_nhwc[0].shape[3]
    keep, a = _feat_args(dfeats_nhwc, scales)
    if dout2 is None:
        L.call("omni_roi_align_bwd", *a, _lib.ptr(rois), _lib.ptr(batch_idx), _lib.ptr(levels), R, P, C, _lib.ptr(dout),
               _lib.stream_of(rois))
        return
    assert dout2.is_contiguous() and (dout is None or dout.is_contiguous())
    L.call("omni_roi_align_bwd2", *a, _lib.ptr(rois), _lib.ptr(batch_idx), _lib.ptr(levels), R, P, C, _lib.ptr(dout), _lib.ptr(dout2),
           int(per_image), int(first), _lib.stream_of(rois))


def box_loss_fwd(pred, K, cls, prop, gt, gt_row, weights=(10.0, 10.0, 5.0, 5.0)):
    L = _dev(pred, cls, prop, gt, gt_row)
    R, ldp = pred.shape
    sums = _empty((7,), torch.float64, pred)
    L.call("omni_box_loss_fwd", _lib.ptr(pred), ldp, R, K, _lib.ptr(cls), _lib.ptr(prop), _lib.ptr(gt), _lib.ptr(gt_row),
           *[float(w) for w in weights], _lib.ptr(sums), _lib.stream_of(pred))
    return sums


def box_loss_bwd(pred, K, cls, prop, gt, gt_row, sums, g_cls, g_reg, weights=(10.0, 10.0, 5.0, 5.0)):
    L = _dev(pred, cls, prop, gt, gt_row, sums, g_cls, g_reg)
    R, ldp = pred.shape
    dpred = torch.empty_like(pred)
    L.call("omni_box_loss_bwd", _lib.ptr(pred), ldp, R, K, _lib.ptr(cls), _lib.ptr(prop), _lib.ptr(gt), _lib.ptr(gt_row),
           *[float(w) for w in weights], _lib.ptr(sums), _lib.ptr(g_cls), _lib.ptr(g_reg), _lib.ptr(dpred),
           _lib.stream_of(pred))
    return dpred


def box_decode_gt_class(pred, K, cls, prop, weights=(10.0, 10.0, 5.0, 5.0), scale_clamp=4.135166556742356):
    """predict_boxes_for_gt_classes (detectron2 FastRCNNOutputLayers; roi_heads.py:276-289): (R,4) predicted box of each row's GT class"""
    L = _dev(pred, cls, prop)
    R, ldp = pred.shape
    out = _empty((R, 4), torch.float32, pred)
    L.call("omni_box_decode_gt_class", _lib.ptr(pred), ldp, R, K, _lib.ptr(cls), _lib.ptr(prop), *[float(w) for w in weights],
           float(scale_clamp), _lib.ptr(out), _lib.stream_of(pred))
    return out


CUBE_MODE_BASE = 0xDC0      # configs/Base.yaml: z direct, dims priors 'exp', 6d pose, allocentric, virtual depth, chamfer, confidence, joint
_Z_TYPES = {"direct": 0, "sigmoid": 1, "log": 2, "clusters": 3}
_POSE_TYPES = {"6d": 0, "quaternion": 1, "euler": 2}
POSE_WIDTH = {"6d": 6, "quaternion": 4, "euler": 3}


def cube_mode(z_type="direct", dims_priors_enabled=True, dims_priors_func="exp", pose_type="6d", allocentric_pose=True,
              virtual_depth=True, chamfer_pose=True, inverse_z_weight=False, use_confidence=True, joint=True, disentangled=True):
    """MODEL.ROI_CUBE_HEAD.* -> the `mode` word of csrc/cube_head.hip (bit layout in include/omni3d_hip.h)."""
    if z_type not in _Z_TYPES:
        raise ValueError(f"MODEL.ROI_CUBE_HEAD.Z_TYPE '{z_type}' is not one of {sorted(_Z_TYPES)}")
    if pose_type not in _POSE_TYPES:
        raise ValueError("Cuboid pose type {} is not recognized".format(pose_type))
    if dims_priors_enabled and dims_priors_func not in ("exp", "sigmoid"):
        raise NotImplementedError(f"DIMS_PRIORS_FUNC '{dims_priors_func}'")
    if not disentangled and dims_priors_enabled:
        raise ValueError("DISENTANGLED_LOSS False needs DIMS_PRIORS_ENABLED False: the reference's entangled dimension loss divides "
                         "an (n,3) tensor by the (n,2,3) priors (roi_heads.py:620-622) and cannot be evaluated")
    dims = 2 if not dims_priors_enabled else (1 if dims_priors_func == "sigmoid" else 0)
    return (_Z_TYPES[z_type] | dims << 2 | _POSE_TYPES[pose_type] << 4 | bool(allocentric_pose) << 6 | bool(virtual_depth) << 7
            | bool(chamfer_pose) << 8 | bool(inverse_z_weight) << 9 | bool(use_confidence) << 10 | bool(joint) << 11
            | (not disentangled) << 12)


def cube_head_width(mode, bins=1):
    """columns per class of the fused head output: xy 2 + z bins + dims 3 + pose 6/4/3 + uncertainty 0/1"""
    return 5 + bins + (6, 4, 3)[(mode >> 4) & 3] + ((mode >> 10) & 1)


def _clusters(clusters):
    """clusters = None or (bins, zscales (K, bins), zstats (K, bins, 2) or None) -> (bins, zscales, zstats)"""
    if clusters is None:
        return 1, None, None
    bins, zscales, zstats = clusters
    return int(bins), zscales, zstats


def cube_loss_fwd(head, K, boxes, cls, img, Ks, v2r, priors, gt3d, gtpose, gt_row, loss_w=(1.0, 1.0, 1.0, 1.0, 1.0), mode=CUBE_MODE_BASE,
                  clusters=None):
    """loss_w = (w_dims, w_pose, w_xy, w_z, w_joint): only the logged total uses them (the loss terms are weighted by the caller)"""
    bins, zscales, zstats = _clusters(clusters)
    L = _dev(head, boxes, cls, img, Ks, v2r, priors, gt3d, gtpose, gt_row, zscales, zstats)
    F, ldh = head.shape
    vals = _empty((max(F, 1), 13), torch.float32, head)
    jac = _empty((max(F, 1), 6, 13), torch.float32, head)
    red = _empty((24,), torch.float32, head)
    L.call("omni_cube_loss_fwd", _lib.ptr(head), ldh, F, K, int(mode), bins, _lib.ptr(zscales), _lib.ptr(zstats), _lib.ptr(boxes),
           _lib.ptr(cls), _lib.ptr(img), _lib.ptr(Ks),
           _lib.ptr(v2r), _lib.ptr(priors), _lib.ptr(gt3d), _lib.ptr(gtpose), _lib.ptr(gt_row), *[float(w) for w in loss_w],
           _lib.ptr(vals), _lib.ptr(jac), _lib.ptr(red), _lib.stream_of(head))
    return vals, jac, red


def cube_loss_bwd(vals, jac, red, gk, cls, boxes, F, K, ldh, mode=CUBE_MODE_BASE, clusters=None):
    bins, zscales, _ = _clusters(clusters)
    L = _dev(vals, jac, red, gk, cls, boxes, zscales)
    dhead = _empty((F, ldh), torch.float32, vals)
    L.call("omni_cube_loss_bwd", _lib.ptr(vals), _lib.ptr(jac), _lib.ptr(red), _lib.ptr(gk), _lib.ptr(cls), _lib.ptr(boxes), F, K,
           int(mode), bins, _lib.ptr(zscales), ldh, _lib.ptr(dhead), _lib.stream_of(vals))
    return dhead


def cube_decode(head, K, boxes, cls, img, Ks, v2r, ratio, priors, mode=CUBE_MODE_BASE, clusters=None):
    bins, zscales, zstats = _clusters(clusters)
    L = _dev(head, boxes, cls, img, Ks, v2r, ratio, priors, zscales, zstats)
    F, ldh = head.shape
    cube3d = _empty((F, 9), torch.float32, head)
    pose = _empty((F, 3, 3), torch.float32, head)
    verts = _empty((F, 8, 3), torch.float32, head)
    L.call("omni_cube_decode", _lib.ptr(head), ldh, F, K, int(mode), bins, _lib.ptr(zscales), _lib.ptr(zstats), _lib.ptr(boxes),
           _lib.ptr(cls), _lib.ptr(img), _lib.ptr(Ks),
           _lib.ptr(v2r), _lib.ptr(ratio), _lib.ptr(priors), _lib.ptr(cube3d), _lib.ptr(pose), _lib.ptr(verts),
           _lib.stream_of(head))
    return cube3d, pose, verts


def cuboid_corners(box3d, R):
    box3d, R = box3d.contiguous().float(), R.contiguous().float()
    L = _dev(box3d, R)
    n = box3d.shape[0]
    verts = _empty((n, 8, 3), torch.float32, box3d)
    L.call("omni_cuboid_corners", _lib.ptr(box3d), _lib.ptr(R), n, _lib.ptr(verts), _lib.stream_of(box3d))
    return verts


def sgd_step(param, grad, buf, lr, momentum=0.9, dampening=0.0, weight_decay=0.0, nesterov=False, first_step=False,
             skip_flag=None, grad_scale=1.0):
    L = _dev(param, grad, buf, skip_flag)
    L.call("omni_sgd_step", _lib.ptr(param), _lib.ptr(grad), _lib.ptr(buf), param.numel(), float(lr), float(momentum),
           float(dampening), float(weight_decay), int(nesterov), int(first_step), float(grad_scale), _lib.ptr(skip_flag),
           _lib.stream_of(param))


def adam_step(param, grad, exp_avg, exp_avg_sq, max_exp_avg_sq, lr, beta1, beta2, eps, weight_decay, decoupled, step, skip_flag=None,
              grad_scale=1.0):
    """torch.optim.Adam (decoupled False) / AdamW (True) over one contiguous range; step: device float, see adam_tick"""
    L = _dev(param, grad, exp_avg, exp_avg_sq, max_exp_avg_sq, step, skip_flag)
    L.call("omni_adam_step", _lib.ptr(param), _lib.ptr(grad), _lib.ptr(exp_avg), _lib.ptr(exp_avg_sq), _lib.ptr(max_exp_avg_sq),
           param.numel(), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(decoupled), _lib.ptr(step),
           float(grad_scale), _lib.ptr(skip_flag), _lib.stream_of(param))


def adam_tick(step, skip_flag=None):
    L = _dev(step, skip_flag)
    L.call("omni_adam_tick", _lib.ptr(step), _lib.ptr(skip_flag), _lib.stream_of(step))


def nonfinite_any(grad, flag):
    L = _dev(grad, flag)
    L.call("omni_nonfinite_any", _lib.ptr(grad), grad.numel(), _lib.ptr(flag), _lib.stream_of(grad))


# ---- batched inference (csrc/infer.hip) ---------------------------------------------------------------------------------
DET_MAX_CANDIDATES = 8192      # candidates per image that enter NMS (the top-k / NMS kernels' capacity).  The reference has no cap;
                               # with trained weights a few hundred (roi, class) pairs pass SCORE_THRESH_TEST


def det_scores(pred, rois, count, image_hw, B, P, K, weights, score_thresh):
    L = _dev(pred, rois, count, image_hw)
    scores = _empty((B, P * K), torch.float32, pred)
    probs = _empty((B * P, K), torch.float32, pred)
    boxes = _empty((B * P * K, 4), torch.float32, pred)
    wx, wy, ww, wh = [float(v) for v in weights]
    L.call("omni_det_scores", _lib.ptr(pred), pred.shape[1], _lib.ptr(rois), _lib.ptr(count), _lib.ptr(image_hw), B, P, K, wx, wy, ww, wh,
           float(score_thresh), _lib.ptr(scores), _lib.ptr(probs), _lib.ptr(boxes), _lib.stream_of(pred))
    return scores, probs, boxes


def det_nms_boxes(boxes, vals, idx, B, PK, K, cap):
    L = _dev(boxes, vals, idx)
    out = _empty((B, cap, 4), torch.float32, boxes)
    valid = _empty((B, cap), torch.int32, boxes)
    L.call("omni_det_nms_boxes", _lib.ptr(boxes), _lib.ptr(vals), _lib.ptr(idx), B, PK, K, cap, _lib.ptr(out), _lib.ptr(valid),
           _lib.stream_of(boxes))
    return out, valid


def det_compact(keep, valid, vals, idx, boxes, B, PK, K, cap, topk):
    L = _dev(keep, valid, vals, idx, boxes)
    obox = _empty((B, topk, 4), torch.float32, boxes)
    oscore = _empty((B, topk), torch.float32, boxes)
    ocls = _empty((B, topk), torch.int32, boxes)
    oroi = _empty((B, topk), torch.int32, boxes)
    ocount = _empty((B,), torch.int32, boxes)
    L.call("omni_det_compact", _lib.ptr(keep), _lib.ptr(valid), _lib.ptr(vals), _lib.ptr(idx), _lib.ptr(boxes), B, PK, K, cap, topk,
           _lib.ptr(obox), _lib.ptr(oscore), _lib.ptr(ocls), _lib.ptr(oroi), _lib.ptr(ocount), _lib.stream_of(boxes))
    return obox, oscore, ocls, oroi, ocount


def guard_pre(vec, n):
    L = _dev(vec)
    L.call("omni_guard_pre", _lib.ptr(vec), n, _lib.stream_of(vec))


def guard_post(vec, n, world, stabilize, half_period, tolerance, gamma, state, skip, out):
    L = _dev(vec, state, skip, out)
    L.call("omni_guard_post", _lib.ptr(vec), n, world, float(stabilize), float(half_period), float(tolerance), float(gamma), _lib.ptr(state),
           _lib.ptr(skip), _lib.ptr(out), _lib.stream_of(vec))
