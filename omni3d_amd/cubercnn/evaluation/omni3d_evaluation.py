"""`box3d_overlap` (reference cubercnn/evaluation/omni3d_evaluation.py:106-166) on the IoU3D kernel, and the batched
form the evaluator needs: `Omni3Deval.evaluate` calls `computeIoU` once per (image, category) group from a Python dict
comprehension (:1339-1343, :1357-1431), i.e. thousands of tiny `box3d_overlap` calls; `box3d_overlap_groups` takes all
groups at once -- one validity launch + one ragged pairs launch + one readback.  The COCO-style greedy matching /
accumulation around it (evaluateImg / accumulate) stays SURVEY.md 8(f) "next"."""
import numpy as np
import torch

from ...kernels import iou3d


def box3d_overlap(boxes_dt: torch.Tensor, boxes_gt: torch.Tensor, eps_coplanar: float = 1e-4, eps_nonzero: float = 1e-8) -> torch.Tensor:
    """(N,8,3), (M,8,3) corner lists -> (N,M) IoU; rows of non-coplanar / zero-area detections are 0."""
    dev = boxes_dt.device
    if not boxes_dt.is_cuda:   # the reference forces this path to the CPU (MAX_DTS_CROSS_GTS_FOR_IOU3D = 0); here it is a GPU op
        dev = torch.device("cuda")
    out = iou3d.box3d_overlap(boxes_dt.to(dev).float(), boxes_gt.to(dev).float(), eps_coplanar, eps_nonzero)
    return out.to(boxes_dt.device)


def box3d_overlap_groups(boxes_dt, boxes_gt, dt_sizes, gt_sizes, eps_coplanar: float = 1e-4, eps_nonzero: float = 1e-8, warn=False):
    """All (image, category) groups of an evaluation in one pass.

    boxes_dt (sum(dt_sizes), 8, 3) / boxes_gt (sum(gt_sizes), 8, 3): the groups' detection (already score-sorted and cut to
    maxDets, :1374-1377) and ground-truth corner lists, concatenated in group order; dt_sizes / gt_sizes: per-group counts.
    -> list of (Nd_g, Ng_g) float32 IoU matrices (views of one flat device tensor, in group order; empty groups give
    empty matrices), each equal to `box3d_overlap(dt_g, gt_g)` of the reference."""
    dt_sizes = np.asarray(dt_sizes, dtype=np.int64)
    gt_sizes = np.asarray(gt_sizes, dtype=np.int64)
    if dt_sizes.shape != gt_sizes.shape or dt_sizes.ndim != 1:
        raise ValueError("dt_sizes / gt_sizes must be 1-D and of equal length")
    if int(dt_sizes.sum()) != boxes_dt.shape[0] or int(gt_sizes.sum()) != boxes_gt.shape[0]:
        raise ValueError("group sizes do not add up to the number of boxes")
    dev = boxes_dt.device if boxes_dt.is_cuda else torch.device("cuda")
    if not boxes_dt.is_cuda and iou3d._lib.get().emulated:      # host-emulated kernels in the GPU-less test suite
        dev = boxes_dt.device
    dt, gt = boxes_dt.to(dev).float().contiguous(), boxes_gt.to(dev).float().contiguous()
    counts = dt_sizes * gt_sizes
    pair_off = np.concatenate([[0], np.cumsum(counts)])
    P = int(pair_off[-1])
    # ragged pair list, row-major inside each group (host index arithmetic on the group table, one upload)
    gid = np.repeat(np.arange(len(counts)), counts)
    local = np.arange(P) - pair_off[gid]
    ng = np.maximum(gt_sizes[gid], 1)
    dt_off = np.concatenate([[0], np.cumsum(dt_sizes)])[:-1]
    gt_off = np.concatenate([[0], np.cumsum(gt_sizes)])[:-1]
    idx1 = torch.from_numpy((dt_off[gid] + local // ng).astype(np.int32)).to(dev)
    idx2 = torch.from_numpy((gt_off[gid] + local % ng).astype(np.int32)).to(dev)
    if P == 0:
        flat = torch.zeros(0, dtype=torch.float32, device=dev)
    else:
        valid, vcounts = iou3d.box3d_validity(dt, eps_coplanar, eps_nonzero)
        _, flat = iou3d.iou_box3d_pairs(dt, gt, idx1, idx2, valid1=valid)
        if warn:
            c = vcounts.tolist()
            if c[0] > 0:
                print('Warning: skipping {:d} non-coplanar boxes at eval.'.format(int(c[0])))
            if c[1] > 0:
                print('Warning: skipping {:d} zero volume boxes at eval.'.format(int(c[1])))
    return [flat[pair_off[g]:pair_off[g + 1]].view(int(dt_sizes[g]), int(gt_sizes[g])) for g in range(len(counts))]


def evaluate_groups(ious_flat, dt_sizes, gt_sizes, gt_ignore, gt_range, dt_range, area_ranges, iou_thrs):
    """The greedy matching of `Omni3Deval.evaluateImg` (:1433-1551, 3D mode, eval_prox off) for every (image, category)
    group x depth range x IoU threshold in one launch (the reference loops over them in Python, :1346-1351).

    ious_flat: the concatenated (D_g, G_g) matrices (e.g. torch.cat of box3d_overlap_groups' views), detections in descending
    score order and cut to maxDets; dt_sizes / gt_sizes: per-group counts; gt_ignore (sumG) int `ignore3D`; gt_range (sumG),
    dt_range (sumD) float `depth`; area_ranges (A, 2); iou_thrs (T,).
    -> dict of device tensors: dt_match (A,T,sumD) index of the matched gt inside its group (original order) or -1,
       gt_match (A,T,sumG) index of the matched dt or -1, dt_ignore (A,T,sumD) uint8, gt_order (A,sumG) the stable
       ignore-last order, gt_ignore (A,sumG) uint8 `_ignore` per original gt."""
    L = iou3d._lib.get()
    dev = ious_flat.device
    dt_sizes, gt_sizes = np.asarray(dt_sizes, dtype=np.int64), np.asarray(gt_sizes, dtype=np.int64)
    ng, sumD, sumG = len(dt_sizes), int(dt_sizes.sum()), int(gt_sizes.sum())
    A, T = len(area_ranges), len(iou_thrs)
    i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)      # noqa: E731
    iou_off = torch.from_numpy(np.concatenate([[0], np.cumsum(dt_sizes * gt_sizes)])[:-1].astype(np.int64)).to(dev)
    dt_off, gt_off = i32(np.concatenate([[0], np.cumsum(dt_sizes)])), i32(np.concatenate([[0], np.cumsum(gt_sizes)]))
    areas = torch.tensor(np.asarray(area_ranges, dtype=np.float32)).to(dev).contiguous()
    thrs = torch.tensor(np.asarray(iou_thrs, dtype=np.float64)).to(dev).contiguous()
    out = {"dt_match": torch.empty((A, T, sumD), dtype=torch.int32, device=dev), "gt_match": torch.empty((A, T, sumG), dtype=torch.int32, device=dev),
           "dt_ignore": torch.empty((A, T, sumD), dtype=torch.uint8, device=dev), "gt_order": torch.empty((A, sumG), dtype=torch.int32, device=dev),
           "gt_ignore": torch.empty((A, sumG), dtype=torch.uint8, device=dev)}
    if ng == 0:
        return out
    ious_flat = ious_flat.float().contiguous()
    gt_ignore, gt_range, dt_range = gt_ignore.to(torch.int32).contiguous(), gt_range.float().contiguous(), dt_range.float().contiguous()
    iou3d._lib.check_device(ious_flat, gt_ignore, gt_range, dt_range)
    _p = iou3d._lib.ptr
    L.call("omni_eval_match", _p(ious_flat), _p(iou_off), _p(dt_off), _p(gt_off), _p(gt_ignore), _p(gt_range), _p(dt_range), _p(areas),
           _p(thrs), ng, A, T, sumD, sumG, int(gt_sizes.max()) if ng else 0, _p(out["dt_match"]), _p(out["gt_match"]),
           _p(out["dt_ignore"]), _p(out["gt_order"]), _p(out["gt_ignore"]), iou3d._lib.stream_of(ious_flat))
    return out
