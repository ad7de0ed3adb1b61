"""state-dict / checkpoint compatibility with the reference's parameter names and shapes
(Appendix C), so released model-zoo `.pth` files (MODEL_ZOO.md) load into the HIP model and vice versa."""
import io
import os

import pytest
import torch


def _model(config):
    from oracle import make_golden as MG
    from omni3d_amd import synthetic
    priors = synthetic.make_priors(50)
    return MG.build_product_model(MG.product_cfg([], config), priors, 0), priors


@pytest.mark.parametrize("config,probe", [
    ("cubercnn_DLA34_FPN.yaml", ["backbone.bottom_up.level2.tree1.conv1.weight", "backbone.bottom_up.level3.tree1.tree1.bn2.running_var",
                                 "backbone.bottom_up.base_layer.1.num_batches_tracked"]),
    ("cubercnn_ResNet34_FPN.yaml", ["backbone.bottom_up.layer2.0.downsample.0.weight", "backbone.bottom_up.bn1.running_mean"]),
    ("cubercnn_densenet_FPN.yaml", ["backbone.bottom_up.base.denseblock4.denselayer16.conv2.weight", "backbone.bottom_up.base.norm5.running_var",
                                    "backbone.bottom_up.base.transition2.conv.weight"]),
    ("cubercnn_mnasnet_FPN.yaml", ["backbone.bottom_up.base.8.0.layers.3.weight", "backbone.bottom_up.base.15.running_mean"]),
    ("cubercnn_shufflenet_FPN.yaml", ["backbone.bottom_up.stage2.0.branch1.0.weight", "backbone.bottom_up.conv5.0.weight"]),
])
def test_state_dict_names_shapes_roundtrip(config, probe):
    model, priors = _model(config)
    sd = model.state_dict()
    for k in probe + ["backbone.fpn_lateral2.weight", "backbone.fpn_output6.bias", "proposal_generator.rpn_head.conv.weight",
                      "proposal_generator.rpn_head.objectness_logits.bias", "proposal_generator.rpn_head.anchor_deltas.weight",
                      "roi_heads.box_head.fc1.weight", "roi_heads.box_predictor.cls_score.weight", "roi_heads.box_predictor.bbox_pred.bias",
                      "roi_heads.cube_head.feature_generator.fc1.weight", "roi_heads.cube_head.bbox_3D_pose.weight",
                      "roi_heads.priors_dims_per_cat", "roi_heads.priors_z_scales"]:
        assert k in sd, k
    # reference shapes (not the kernel-side layouts): fc1 is (1024, 12544), RPN heads are three separate convs
    assert tuple(sd["roi_heads.box_head.fc1.weight"].shape) == (1024, 12544)
    assert tuple(sd["roi_heads.cube_head.feature_generator.fc1.weight"].shape) == (1024, 12544)
    assert tuple(sd["proposal_generator.rpn_head.objectness_logits.weight"].shape) == (3, 256, 1, 1)
    assert tuple(sd["proposal_generator.rpn_head.anchor_deltas.weight"].shape) == (12, 256, 1, 1)
    assert tuple(sd["roi_heads.priors_dims_per_cat"].shape) == (1, 50, 2, 3)
    n_params = sum(p.numel() for p in model.parameters())
    if "DLA34" in config:
        assert n_params == 47_908_514          # SURVEY.md Appendix C / the reference's RCNN3D, strict-loaded in make_golden
    # file round trip through torch.save (what DetectionCheckpointer / PeriodicCheckpointerOnlyOne write: {"model": sd, ...})
    buf = io.BytesIO()
    torch.save({"model": sd, "iteration": 7}, buf)
    buf.seek(0)
    ck = torch.load(buf, weights_only=False)
    fresh, _ = _model(config)
    with torch.no_grad():
        for p in fresh.parameters():
            p.add_(1.0)
    missing = fresh.load_state_dict(ck["model"], strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    for (k, a), (_, b) in zip(sorted(fresh.state_dict().items()), sorted(sd.items())):
        assert torch.equal(a, b), k


def test_periodic_checkpointer_only_one(tmp_path):
    from omni3d_amd.cubercnn.solver.checkpoint import PeriodicCheckpointerOnlyOne

    class Ck:
        def __init__(self):
            self.saved = []

        def save(self, name, **kw):
            self.saved.append((name, kw["iteration"]))
    ck = Ck()
    pc = PeriodicCheckpointerOnlyOne(ck, period=3, max_iter=8)
    for it in range(8):
        pc.step(it)
    assert ck.saved == [("model_recent", 2), ("model_recent", 5), ("model_final", 7)]
