"""Eval-mode parity (RCNN3D.inference, /root/reference/cubercnn/modeling/meta_arch/rcnn3d.py:79-112;
fast_rcnn_inference_single_image fast_rcnn.py:57-116; cube eval branch roi_heads.py:353-357,771-824) against
the fixture produced by the reference's own files (oracle/make_golden.py --infer)."""
import os

import pytest
import torch

from conftest import ROOT

GOLD = os.path.join(ROOT, "tests", "golden", "dla34_small_infer.pt")


def _run(dev):
    from oracle import make_golden as MG
    from omni3d_amd import synthetic
    gold = torch.load(GOLD, weights_only=False)
    spec = gold["spec"]
    priors = synthetic.make_priors(50)
    model = MG.sharpen(MG.build_product_model(MG.product_cfg(spec["overrides"]), priors, spec["seed"])).to(dev)
    batch = synthetic.make_batch(spec["images"], spec["height"], spec["width"], num_gt=spec["num_gt"], seed=spec["seed"], priors=priors)
    for b in batch:
        b.pop("instances")
        b["height"], b["width"] = 2 * spec["height"], 2 * spec["width"]
        b["K"] = [[2 * v for v in row] for row in b["K"][:2]] + [b["K"][2]]
    model.eval()
    with torch.no_grad():
        out = model(batch)
    assert len(out) == len(gold["results"])
    for o, ref in zip(out, gold["results"]):
        i = o["instances"]
        n = len(ref["scores"])
        assert len(i) == n, (len(i), n)
        assert torch.equal(i.pred_classes.cpu().long(), ref["pred_classes"])            # selection is index-exact
        def close(a, b, tol):   # fp32 bar of the north star (1e-4), relative for values above 1
            err = float(((a.cpu() - b).abs() / (1.0 + b.abs())).max()) if b.numel() else 0.0
            assert err <= tol, (err, tol)
            return True
        assert close(i.scores, ref["scores"], 1e-4)
        px = 1e-4 * max(o["instances"].image_size)     # pixel quantities: 1e-4 of the image extent
        assert (i.pred_boxes.tensor.cpu() - ref["pred_boxes"]).abs().max() < px
        # dims = prior * exp(logit), z = exp-style depth decode: the fp32 summation order of ~60 conv layers + two 12544-long
        # GEMM reductions (MFMA k-slab order vs the CPU's) shows up as 1.3e-4 relative here; CPU fp32 in a different memory
        # format moves the same amount against itself (DESIGN.md "conditioning"), so the bar is 3e-4
        assert close(i.pred_dimensions, ref["pred_dimensions"], 3e-4)
        assert close(i.pred_center_cam, ref["pred_center_cam"], 3e-4)
        c2 = ref["pred_center_2D"]          # projected 3D centres, may lie far outside the image (|u| up to ~800 px here)
        assert bool(((i.pred_center_2D.cpu() - c2).abs() <= 1e-4 * c2.abs().clamp(min=max(o["instances"].image_size)) + 0.01).all())
        # the Gram-Schmidt of a random-init 6D pose (|a| ~ 1e-2, a1 and a2 far from orthogonal) amplifies fp32
        # rounding of the head GEMMs: rotation entries and the corners built from them get a looser bar
        assert close(i.pred_pose, ref["pred_pose"], 2e-3)
        assert close(i.pred_bbox3D, ref["pred_bbox3D"], 2e-3)


@pytest.mark.skipif(os.environ.get("OMNI_SLOW") != "1", reason="minutes under the host emulator; set OMNI_SLOW=1 (the GPU variant is the gate)")
def test_inference_matches_reference_emulated(emu_lib):
    _run("cpu")


@pytest.mark.gpu
def test_inference_matches_reference_gpu(hip_lib, deterministic_forward):
    _run("cuda")
