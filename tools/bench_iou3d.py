#!/usr/bin/env python
"""A/B of the IoU3D kernel's lanes-per-pair width on the bench workload (100k Omni3D-shaped pairs, BASELINE configs[4]):
HIP-event time per launch -> pairs/s for 64 (one pair per wave), 32 (two) and 16 (four) lanes per pair."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from omni3d_amd import boxgen  # noqa: E402
from omni3d_amd.kernels import iou3d  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
dt, gt, _ = boxgen.omni3d_like_pairs(np.random.default_rng(1000), P)
d, g = torch.from_numpy(dt).cuda(), torch.from_numpy(gt).cuda()
ar = torch.arange(P, dtype=torch.int32, device="cuda")
valid, _ = iou3d.box3d_validity(d)
ref = None
for lanes in (64, 32, 16, 1064, 1032, 1016, 2064, 2032, 3032):
    for _ in range(3):
        out = iou3d.iou_box3d_pairs(d, g, ar, ar, valid1=valid, lanes_per_pair=lanes)[1]
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(20):
        out = iou3d.iou_box3d_pairs(d, g, ar, ar, valid1=valid, lanes_per_pair=lanes)[1]
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    ref = out if ref is None else ref
    print(f"variant {lanes:4d}: {ms:.3f} ms / launch = {P / ms * 1e3:.3e} pairs/s   max|d| vs 64-lane {float((out - ref).abs().max()):.2e}")
