"""`pack_instances_cached` must never hand the previous step's ground truth / intrinsics to a NEW list that
happens to live at a recycled address (CPython reuses the address of a dead list immediately), nor reuse intrinsics that were
not compared.  Host logic only."""
import torch

from conftest import ROOT  # noqa: F401


def _inst(x0, cls):
    from omni3d_amd.d2.structures import Boxes, Instances
    i = Instances((64, 64))
    i.gt_boxes = Boxes(torch.tensor([[x0, 2.0, x0 + 20.0, 30.0]]))
    i.gt_classes = torch.tensor([cls])
    return i


def test_new_list_at_a_recycled_address_is_repacked():
    from omni3d_amd.cubercnn.modeling.targets import pack_instances_cached
    sizes = [(64, 64), (64, 64)]
    seen_ids = set()
    for step in range(6):
        gt = [_inst(float(step), step), _inst(float(step) + 1, step)]          # a fresh list per iteration, like a data loader
        seen_ids.add(id(gt))
        t = pack_instances_cached(gt, sizes)
        assert t.gt_cls.tolist() == [step, step] and float(t.gt[0, 0]) == float(step)
        assert pack_instances_cached(gt, sizes) is t                          # same list object within the step: one packing
        del gt, t
    # (the loop is only a regression if addresses really were recycled; they are in CPython, but do not assert on it)


def test_refilled_list_and_intrinsics_are_compared():
    from omni3d_amd.cubercnn.modeling.targets import pack_instances_cached
    sizes = [(64, 64)]
    gt = [_inst(1.0, 3)]
    K1 = [torch.tensor([[500.0, 0, 32], [0, 500.0, 32], [0, 0, 1]])]
    K2 = [torch.tensor([[250.0, 0, 32], [0, 250.0, 32], [0, 0, 1]])]
    a = pack_instances_cached(gt, sizes)                       # RPN: no intrinsics
    b = pack_instances_cached(gt, sizes, K1, [1.0])            # ROI heads upgrade the entry
    assert not a.has_intrinsics and b.has_intrinsics and float(b.Ks[0, 0]) == 500.0
    assert pack_instances_cached(gt, sizes, K1, [1.0]) is b
    assert pack_instances_cached(gt, sizes) is b               # a later call without intrinsics may use the richer entry
    c = pack_instances_cached(gt, sizes, K2, [1.0])            # same list, other intrinsics: never reuse
    assert c is not b and float(c.Ks[0, 0]) == 250.0
    gt[0] = _inst(9.0, 7)                                      # list refilled in place
    d = pack_instances_cached(gt, sizes, K2, [1.0])
    assert d is not c and d.gt_cls.tolist() == [7]
    # eval: `[None] * B` temporaries are never cached against each other
    e1 = pack_instances_cached([None], sizes, K1, [1.0])
    e2 = pack_instances_cached([None], sizes, K2, [1.0])
    assert float(e1.Ks[0, 0]) == 500.0 and float(e2.Ks[0, 0]) == 250.0
