"""IoU3D launchers.  Mirrors the reference interface
``box3d_overlap(boxes_dt, boxes_gt, eps_coplanar=1e-4, eps_nonzero=1e-8) -> ious``
(/root/reference/cubercnn/evaluation/omni3d_evaluation.py:106-166) and pytorch3d's
``_C.iou_box3d(boxes1, boxes2) -> (vol, iou)`` (call site omni3d_evaluation.py:155).
"""
import torch

from .. import lib as _lib


def _check_boxes(b, name):
    if b.dim() != 3 or b.shape[1] != 8 or b.shape[2] != 3:
        raise ValueError(f"{name} must have shape (B, 8, 3), got {tuple(b.shape)}")
    if b.dtype != torch.float32:
        raise ValueError(f"{name} must be float32")


def iou_box3d(boxes1, boxes2, valid1=None):
    """(N,8,3),(M,8,3) -> (vol (N,M), iou (N,M)); optional int32 mask of valid rows."""
    _check_boxes(boxes1, "boxes1")
    _check_boxes(boxes2, "boxes2")
    boxes1, boxes2 = boxes1.contiguous(), boxes2.contiguous()
    L = _lib.check_device(boxes1, boxes2, valid1)
    N, M = boxes1.shape[0], boxes2.shape[0]
    vol = torch.empty((N, M), dtype=torch.float32, device=boxes1.device)
    iou = torch.empty((N, M), dtype=torch.float32, device=boxes1.device)
    overflow = torch.zeros(1, dtype=torch.int32, device=boxes1.device)
    L.call("omni_iou_box3d", _lib.ptr(boxes1), N, _lib.ptr(boxes2), M, _lib.ptr(valid1), _lib.ptr(vol), _lib.ptr(iou),
           _lib.ptr(overflow), _lib.stream_of(boxes1))
    return vol, iou


def iou_box3d_pairs(boxes1, boxes2, idx1, idx2, valid1=None, lanes_per_pair=0):
    """Ragged / paired form: iou[p] = IoU3D(boxes1[idx1[p]], boxes2[idx2[p]]).  lanes_per_pair: launch variant of
    omni_iou_box3d_pairs_algo (64 / 32 / 16 lanes per pair, + 1000 = small LDS lists with a retry pass; 0 = production choice);
    every variant gives the same result."""
    _check_boxes(boxes1, "boxes1")
    _check_boxes(boxes2, "boxes2")
    boxes1, boxes2 = boxes1.contiguous(), boxes2.contiguous()
    idx1 = idx1.to(torch.int32).contiguous()
    idx2 = idx2.to(torch.int32).contiguous()
    if idx1.shape != idx2.shape or idx1.dim() != 1:
        raise ValueError("idx1/idx2 must be 1-D and of equal length")
    L = _lib.check_device(boxes1, boxes2, idx1, idx2, valid1)
    P = idx1.numel()
    vol = torch.empty(P, dtype=torch.float32, device=boxes1.device)
    iou = torch.empty(P, dtype=torch.float32, device=boxes1.device)
    overflow = torch.zeros(1, dtype=torch.int32, device=boxes1.device)
    if lanes_per_pair:
        L.call("omni_iou_box3d_pairs_algo", _lib.ptr(boxes1), _lib.ptr(boxes2), _lib.ptr(idx1), _lib.ptr(idx2), P,
               _lib.ptr(valid1), _lib.ptr(vol), _lib.ptr(iou), _lib.ptr(overflow), int(lanes_per_pair), _lib.stream_of(boxes1))
    else:
        L.call("omni_iou_box3d_pairs", _lib.ptr(boxes1), _lib.ptr(boxes2), _lib.ptr(idx1), _lib.ptr(idx2), P,
               _lib.ptr(valid1), _lib.ptr(vol), _lib.ptr(iou), _lib.ptr(overflow), _lib.stream_of(boxes1))
    return vol, iou


def box3d_validity(boxes, eps_coplanar=1e-4, eps_nonzero=1e-8):
    """-> (valid int32 (N,), counts int32 (2,) = [#non-coplanar, #zero-area])."""
    _check_boxes(boxes, "boxes")
    boxes = boxes.contiguous()
    L = _lib.check_device(boxes)
    N = boxes.shape[0]
    valid = torch.empty(N, dtype=torch.int32, device=boxes.device)
    counts = torch.zeros(2, dtype=torch.int32, device=boxes.device)
    L.call("omni_box3d_validity", _lib.ptr(boxes), N, float(eps_coplanar), float(eps_nonzero), _lib.ptr(valid),
           _lib.ptr(counts), _lib.stream_of(boxes))
    return valid, counts


def box3d_overlap(boxes_dt, boxes_gt, eps_coplanar=1e-4, eps_nonzero=1e-8, warn=True):
    """Drop-in for the reference's ``box3d_overlap``: (N,8,3),(M,8,3) -> iou (N,M) with the rows
    of non-coplanar / zero-area detection boxes zeroed (and the same warnings printed)."""
    valid, counts = box3d_validity(boxes_dt, eps_coplanar, eps_nonzero)
    _, iou = iou_box3d(boxes_dt, boxes_gt, valid1=valid)
    if warn:
        c = counts.tolist()
        if c[0] > 0:
            print('Warning: skipping {:d} non-coplanar boxes at eval.'.format(int(c[0])))
        if c[1] > 0:
            print('Warning: skipping {:d} zero volume boxes at eval.'.format(int(c[1])))
    return iou
