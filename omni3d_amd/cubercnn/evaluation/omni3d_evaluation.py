"""`box3d_overlap` (reference cubercnn/evaluation/omni3d_evaluation.py:106-166) on the IoU3D kernel.
The COCO-style matching / accumulation around it (Omni3Deval) is SURVEY.md 8(f) "next"."""
import torch

from ...kernels import iou3d


def box3d_overlap(boxes_dt: torch.Tensor, boxes_gt: torch.Tensor, eps_coplanar: float = 1e-4, eps_nonzero: float = 1e-8) -> torch.Tensor:
    """(N,8,3), (M,8,3) corner lists -> (N,M) IoU; rows of non-coplanar / zero-area detections are 0."""
    dev = boxes_dt.device
    if not boxes_dt.is_cuda:   # the reference forces this path to the CPU (MAX_DTS_CROSS_GTS_FOR_IOU3D = 0); here it is a GPU op
        dev = torch.device("cuda")
    out = iou3d.box3d_overlap(boxes_dt.to(dev).float(), boxes_gt.to(dev).float(), eps_coplanar, eps_nonzero)
    return out.to(boxes_dt.device)
