"""`Omni3Deval` evaluate -> accumulate -> summarize on the device, pinned to a fixture written by the
REFERENCE's own class (oracle/make_golden.py --evalfull: omni3d_evaluation.py:1172-1357 run under the harness): the
precision / recall / score tables must be identical (integer decisions, doubles compared exactly up to 1e-12)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT

GOLD = os.path.join(ROOT, "tests", "golden", "eval_full.npz")


def _run():
    from omni3d_amd.cubercnn.evaluation import AnnotationIndex, Omni3Deval
    z = np.load(GOLD)
    ann = json.loads(bytes(z["annotations"]).decode())
    ev = Omni3Deval(AnnotationIndex(ann["gts"], range(ann["images"]), range(ann["cats"])),
                    AnnotationIndex(ann["dts"], range(ann["images"]), range(ann["cats"])), mode="3D")
    ev.evaluate()
    ev.accumulate()
    log = ev.summarize()
    for k in ("precision", "recall", "scores"):
        got, want = ev.eval[k], z[k]
        assert got.shape == want.shape, (k, got.shape, want.shape)
        assert np.array_equal(got == -1, want == -1), k                    # absent categories / ranges stay -1
        assert np.abs(got - want).max() <= 1e-12, (k, float(np.abs(got - want).max()), int(np.abs(got - want).argmax()))
    assert np.abs(ev.stats - z["stats"]).max() <= 1e-12
    assert log.count("\n") == 12 and "Average Precision" in log
    # the per-(category, range, image) records of the reference layout, materialised from the device results
    assert len(ev.evalImgs) == ann["cats"] * 4 * ann["images"]
    return ev


def test_omni3deval_matches_reference_fixture_emulated(emu_lib):
    _run()


@pytest.mark.gpu
def test_omni3deval_matches_reference_fixture_gpu(hip_lib):
    _run()


def test_instances_to_coco_json_fields():
    from omni3d_amd.cubercnn.evaluation import instances_to_coco_json
    from omni3d_amd.d2.structures import Boxes, Instances
    inst = Instances((100, 200))
    inst.pred_boxes = Boxes(torch.tensor([[10.0, 20.0, 50.0, 80.0], [0.0, 0.0, 5.0, 5.0]]))
    inst.scores = torch.tensor([0.9, 0.2])
    inst.pred_classes = torch.tensor([3, 7])
    inst.pred_bbox3D = torch.arange(48, dtype=torch.float32).reshape(2, 8, 3)
    inst.pred_center_cam = torch.ones(2, 3)
    inst.pred_center_2D = torch.ones(2, 2)
    inst.pred_dimensions = torch.ones(2, 3)
    inst.pred_pose = torch.eye(3).repeat(2, 1, 1)
    out = instances_to_coco_json(inst, 17)
    assert [o["category_id"] for o in out] == [3, 7] and out[0]["image_id"] == 17
    assert out[0]["bbox"] == [10.0, 20.0, 40.0, 60.0]                                          # XYWH
    assert abs(out[0]["depth"] - float(inst.pred_bbox3D[0, :, 2].mean())) < 1e-6               # mean corner depth (:1002)
    assert set(out[0]) == {"image_id", "category_id", "bbox", "score", "depth", "bbox3D", "center_cam", "center_2D", "dimensions", "pose"}
    assert instances_to_coco_json(Instances((1, 1), pred_boxes=Boxes(torch.zeros(0, 4)), scores=torch.zeros(0), pred_classes=torch.zeros(0)), 0) == []
