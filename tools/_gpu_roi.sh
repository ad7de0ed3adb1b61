#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_det.py tests/test_determinism.py -x -q -m gpu -k "roi_align or bit_identical" 2>&1 | tail -2

cat > /tmp/roi_time.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
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
det.roi_align_bwd_det = wrapped; HF.det.roi_align_bwd_det = wrapped
opt.zero_grad(); total_loss(model(batch, packed)).backward(); torch.cuda.synchronize()
a, kw = rec["args"], rec["kw"]
for _ in range(3): orig(*a, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): orig(*a, **kw)
e1.record(); torch.cuda.synchronize()
print("roi_align_bwd_det (footprint + gather): %.1f us" % (e0.elapsed_time(e1) * 1e3 / 20))
z = [torch.zeros_like(t) for t in a[0]]
e0.record()
for _ in range(20): det.roi_align_bwd(z, *a[1:], **kw)
e1.record(); torch.cuda.synchronize()
print("roi_align_bwd (atomic scatter, without the zero fill): %.1f us" % (e0.elapsed_time(e1) * 1e3 / 20))
PY
python /tmp/roi_time.py 2>&1 | tail -2
