"""DIAGNOSTIC (GPU box): the ROI set the benchmark's training step hands to ROIAlign backward -- per FPN level: ROIs, footprint sizes,
(8 x 8 tile, ROI) overlaps, longest per-tile list -- to size omni_roi_align_bwd_det's work."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from omni3d_amd import bench_train as BT
    from omni3d_amd.kernels import det
    from omni3d_amd.functional import total_loss
    cfg, model, opt, priors = BT.build(1)
    batch, packed = BT.stage_batch(model, priors, 0)
    rec = {}
    orig = det.roi_align_bwd_det

    def wrapped(dfe, scales, rois, bidx, levels, P, dout, **kw):
        rec.update(rois=rois.cpu().numpy(), lv=levels.cpu().numpy(), bidx=bidx.cpu().numpy(), shapes=[tuple(d.shape) for d in dfe], scales=list(scales))
        return orig(dfe, scales, rois, bidx, levels, P, dout, **kw)
    det.roi_align_bwd_det = wrapped
    import omni3d_amd.functional as HF
    HF.det.roi_align_bwd_det = wrapped
    opt.zero_grad()
    total_loss(model(batch, packed)).backward()
    torch.cuda.synchronize()
    rois, lv, bidx = rec["rois"], rec["lv"], rec["bidx"]
    print("ROIs", len(rois), "per level", np.bincount(lv, minlength=4).tolist())
    tot_jobs = 0
    for l, (shape, sc) in enumerate(zip(rec["shapes"], rec["scales"])):
        B, H, W, C = shape
        sel = lv == l
        if not sel.any():
            print(f'level {l}: no ROIs')
            continue
        r = rois[sel] * sc
        w, h = r[:, 2] - r[:, 0], r[:, 3] - r[:, 1]
        x0, x1 = np.floor(np.clip(r[:, 0] - 0.5, 0, W - 1)).astype(int), np.ceil(np.clip(r[:, 2] - 0.5, 0, W - 1)).astype(int)
        y0, y1 = np.floor(np.clip(r[:, 1] - 0.5, 0, H - 1)).astype(int), np.ceil(np.clip(r[:, 3] - 0.5, 0, H - 1)).astype(int)
        tiles = ((x1 // 8 - x0 // 8 + 1) * (y1 // 8 - y0 // 8 + 1))
        cnt = np.zeros((B, (H + 7) // 8, (W + 7) // 8), int)
        for i, n in enumerate(bidx[sel]):
            cnt[n, y0[i] // 8: y1[i] // 8 + 1, x0[i] // 8: x1[i] // 8 + 1] += 1
        tot_jobs += tiles.sum()
        print(f"level {l}: map {H}x{W}  ROIs {sel.sum():5d}  footprint w {np.median(w):5.1f} (max {w.max():5.1f}) h {np.median(h):5.1f} (max {h.max():5.1f})  "
              f"tiles/ROI mean {tiles.mean():5.1f}  (tile,ROI) jobs {tiles.sum():6d}  list length mean {cnt.mean():5.1f} max {cnt.max()}  footprint px {((x1-x0+1)*(y1-y0+1)).sum()}")
    print("total (tile, ROI) jobs", tot_jobs, "x 4 channel groups =", 4 * tot_jobs)


if __name__ == "__main__":
    main()
