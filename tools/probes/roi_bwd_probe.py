"""DIAGNOSTIC (GPU box): capture the ROI set of the benchmark's training step, save it (gpurun_out/roi_bwd_case.pt: rois, levels, image
index, level shapes / scales -- no gradients) and time omni_roi_align_bwd_det on it: whole, per FPN level, and with the ROI order kept
but the list lengths halved (every second ROI) -- to see whether the launch is bound by its longest lists or by its total work."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    from omni3d_amd import bench_train as BT
    from omni3d_amd.kernels import det
    from omni3d_amd.functional import total_loss
    import omni3d_amd.functional as HF
    cfg, model, opt, priors = BT.build(1)
    batch, packed = BT.stage_batch(model, priors, 0)
    rec = {}
    orig = det.roi_align_bwd_det

    def wrapped(dfe, scales, rois, bidx, levels, P, dout, **kw):
        rec.update(args=(dfe, scales, rois, bidx, levels, P, dout), kw=kw)
        return orig(dfe, scales, rois, bidx, levels, P, dout, **kw)
    det.roi_align_bwd_det = wrapped
    HF.det.roi_align_bwd_det = wrapped
    opt.zero_grad()
    total_loss(model(batch, packed)).backward()
    torch.cuda.synchronize()
    dfe, scales, rois, bidx, levels, P, dout = rec["args"]
    kw = rec["kw"]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    torch.save({"rois": rois.cpu(), "bidx": bidx.cpu(), "levels": levels.cpu(), "shapes": [tuple(d.shape) for d in dfe], "scales": list(scales),
                "kw": {k: (v if not torch.is_tensor(v) else tuple(v.shape)) for k, v in kw.items()}}, os.path.join(ROOT, "gpurun_out", "roi_bwd_case.pt"))
    print("kw", {k: (v if not torch.is_tensor(v) else tuple(v.shape)) for k, v in kw.items()}, "dout", None if dout is None else tuple(dout.shape))
    print("as in the step: %.1f us" % timed(lambda: orig(dfe, scales, rois, bidx, levels, P, dout, **kw)))
    if dout is None:
        dout = torch.randn(rois.shape[0], 7, 7, dfe[0].shape[3], device=rois.device)
    print("one gradient tensor: %.1f us" % timed(lambda: orig(dfe, scales, rois, bidx, levels, P, dout)))
    for l in range(len(dfe)):
        sel = (levels == l).nonzero().flatten()
        if sel.numel() == 0:
            continue
        r, b, lv, d = rois[sel].contiguous(), bidx[sel].contiguous(), levels[sel].contiguous(), dout[sel].contiguous()
        print("level %d only (%d ROIs): %.1f us" % (l, sel.numel(), timed(lambda: orig(dfe, scales, r, b, lv, P, d))))
    sel = torch.arange(0, rois.shape[0], 2, device=rois.device)
    r, b, lv, d = rois[sel].contiguous(), bidx[sel].contiguous(), levels[sel].contiguous(), dout[sel].contiguous()
    print("every second ROI: %.1f us" % timed(lambda: orig(dfe, scales, r, b, lv, P, d)))


if __name__ == "__main__":
    main()
