"""DIAGNOSTIC (GPU box): how much of the full-size inference fixture's pose error comes from the fc GEMMs of the ROI heads?
Runs tests/test_inference_parity._run("cuda", "dla34_full_infer") with kernels.conv.linear_fwd (a) as shipped, twice (run-to-run
spread of the atomically split sums), (b) evaluated in float64 by torch.matmul -- NOT a product path, a measuring stick."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def report(tag):
    import test_inference_parity as T
    try:
        T._run("cuda", "dla34_full_infer")
        v = "pass"
    except AssertionError as e:
        v = "FAIL"
    rep = open(os.path.join(ROOT, "gpurun_out", "dla34_full_infer_fp64_report.txt")).read().splitlines()
    pose = [l.split()[2] for l in rep if l.startswith("pred_pose")]
    box = [l.split()[2] for l in rep if l.startswith("pred_bbox3D")]
    dims = [l.split()[2] for l in rep if l.startswith("pred_dimensions")]
    print(f"{tag:40s} {v}  pose {pose}  bbox3D {box}  dims {dims}", flush=True)


def main():
    from omni3d_amd.kernels import conv
    import omni3d_amd.functional as HF
    report("shipped kernels, run 1")
    report("shipped kernels, run 2")
    orig = conv.linear_fwd

    def lin64(x, w, bias=None, relu=False):
        y = x.double() @ w.double().t()
        if bias is not None:
            y = y + bias.double()
        if relu:
            y = y.clamp_(min=0)
        return y.float()
    conv.linear_fwd = lin64
    for mod in list(sys.modules.values()):
        if mod is not None and getattr(mod, "linear_fwd", None) is orig:
            mod.linear_fwd = lin64
    report("every linear layer in float64 (torch)")


if __name__ == "__main__":
    main()
