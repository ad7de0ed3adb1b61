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
