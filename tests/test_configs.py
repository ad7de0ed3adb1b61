"""Every config file of the reference resolves, through this package's own `get_cfg` / `get_cfg_defaults` / YAML presets, to the
same configuration the reference builds from its files (config/config.py + configs/*.yaml + detectron2 defaults as restated in
oracle/upstream.py) -- key for key."""
import os

import pytest

from conftest import ROOT

REF = "/root/reference"
NAMES = ["Base.yaml", "Base_Omni3D.yaml", "Base_Omni3D_in.yaml", "Base_Omni3D_out.yaml", "cubercnn_DLA34_FPN.yaml", "cubercnn_ResNet34_FPN.yaml",
         "cubercnn_densenet_FPN.yaml", "cubercnn_mnasnet_FPN.yaml", "cubercnn_shufflenet_FPN.yaml"]


def _flat(node, prefix=""):
    out = {}
    for k, v in node.items():
        if hasattr(v, "items"):
            out.update(_flat(v, prefix + k + "."))
        else:
            out[prefix + k] = tuple(v) if isinstance(v, (list, tuple)) else v
    return out


def _product(name):
    from omni3d_amd.cubercnn.config import get_cfg_defaults
    from omni3d_amd.d2.config import get_cfg
    cfg = get_cfg()
    get_cfg_defaults(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", name))
    return cfg


def test_every_reference_config_has_a_preset_here():
    assert sorted(os.listdir(os.path.join(ROOT, "configs"))) == sorted(NAMES)
    for n in NAMES:
        _product(n)


@pytest.mark.skipif(not os.path.exists(REF), reason="needs the reference checkout (build container only)")
@pytest.mark.parametrize("name", NAMES)
def test_config_resolves_like_the_reference(name):
    from oracle import ref_harness as H
    from oracle import upstream as U
    H.install()
    from cubercnn.config import get_cfg_defaults
    ref = U.get_cfg()
    get_cfg_defaults(ref)
    ref.merge_from_file(os.path.join(REF, "configs", name))
    a, b = _flat(_product(name)), _flat(ref)
    # every key the reference defines (its own + the upstream defaults the oracle restates) exists here with the same value
    missing = sorted(k for k in b if k not in a)
    assert not missing, missing[:20]
    diff = {k: (a[k], b[k]) for k in b if a[k] != b[k]}
    assert not diff, dict(list(diff.items())[:10])


@pytest.mark.skipif(not os.path.exists(REF), reason="needs the reference checkout (build container only)")
def test_cubercnn_defaults_equal_the_reference_file_key_for_key():
    """The reference's OWN `get_cfg_defaults` (/root/reference/cubercnn/config/config.py:4-158, imported unchanged through the
    harness) and this package's, applied to the same base tree with no YAML on top: the same keys -- in BOTH directions --, the same
    values and the same value types.  (What the harness cannot provide is detectron2's `get_cfg()` itself: that base tree is pinned
    below against the upstream values SURVEY.md Appendix B lists.)"""
    from oracle import ref_harness as H
    H.install()
    from cubercnn.config import get_cfg_defaults as ref_defaults
    from omni3d_amd.cubercnn.config import get_cfg_defaults as prod_defaults
    from omni3d_amd.d2.config import get_cfg
    assert ref_defaults.__code__.co_filename.startswith(REF) and not prod_defaults.__code__.co_filename.startswith(REF)
    base = _flat(get_cfg())
    r, p = get_cfg(), get_cfg()
    ref_defaults(r)
    prod_defaults(p)
    a, b = _flat(p), _flat(r)
    assert sorted(a) == sorted(b), (sorted(set(a) - set(b))[:10], sorted(set(b) - set(a))[:10])
    diff = {k: (a[k], b[k]) for k in b if a[k] != b[k] or type(a[k]) is not type(b[k])}
    assert not diff, dict(list(diff.items())[:10])
    added = [k for k in b if k not in base]
    assert len(added) >= 40, len(added)        # config.py adds the DATASETS / ROI_CUBE_HEAD / DLA / ... keys: they were really compared


# detectron2 v0.6 `config/defaults.py` values the hot path reads (upstream is not installable here, so these
# are literals, independent of omni3d_amd.d2.config)
UPSTREAM_DEFAULTS = {
    "MODEL.DEVICE": "cuda", "INPUT.FORMAT": "BGR", "MODEL.FPN.OUT_CHANNELS": 256, "MODEL.FPN.NORM": "", "MODEL.FPN.FUSE_TYPE": "sum",
    "MODEL.ANCHOR_GENERATOR.NAME": "DefaultAnchorGenerator", "MODEL.ANCHOR_GENERATOR.OFFSET": 0.0,
    "MODEL.RPN.BATCH_SIZE_PER_IMAGE": 256, "MODEL.RPN.IOU_LABELS": (0, -1, 1), "MODEL.RPN.NMS_THRESH": 0.7,
    "MODEL.RPN.BBOX_REG_WEIGHTS": (1.0, 1.0, 1.0, 1.0), "MODEL.RPN.SMOOTH_L1_BETA": 0.0, "MODEL.RPN.BBOX_REG_LOSS_TYPE": "smooth_l1",
    "MODEL.RPN.LOSS_WEIGHT": 1.0, "MODEL.RPN.BBOX_REG_LOSS_WEIGHT": 1.0, "MODEL.RPN.CONV_DIMS": (-1,),
    "MODEL.ROI_HEADS.IOU_THRESHOLDS": (0.5,), "MODEL.ROI_HEADS.IOU_LABELS": (0, 1), "MODEL.ROI_HEADS.POSITIVE_FRACTION": 0.25,
    "MODEL.ROI_HEADS.NMS_THRESH_TEST": 0.5, "MODEL.ROI_HEADS.PROPOSAL_APPEND_GT": True,
    "MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS": (10.0, 10.0, 5.0, 5.0), "MODEL.ROI_BOX_HEAD.SMOOTH_L1_BETA": 0.0,
    "MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO": 0, "MODEL.ROI_BOX_HEAD.POOLER_TYPE": "ROIAlignV2", "MODEL.ROI_BOX_HEAD.FC_DIM": 1024,
    "MODEL.ROI_BOX_HEAD.NUM_CONV": 0, "MODEL.ROI_BOX_HEAD.NORM": "", "MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG": False,
    "MODEL.ROI_BOX_HEAD.TRAIN_ON_PRED_BOXES": False, "MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_TYPE": "smooth_l1",
    "SOLVER.MOMENTUM": 0.9, "SOLVER.NESTEROV": False, "SOLVER.WEIGHT_DECAY_NORM": 0.0, "SOLVER.BIAS_LR_FACTOR": 1.0,
    "SOLVER.WEIGHT_DECAY_BIAS": None, "SOLVER.GAMMA": 0.1, "SOLVER.WARMUP_FACTOR": 0.001, "SOLVER.WARMUP_ITERS": 1000,
    "SOLVER.WARMUP_METHOD": "linear", "SOLVER.CHECKPOINT_PERIOD": 5000, "SOLVER.CLIP_GRADIENTS.ENABLED": False, "SOLVER.AMP.ENABLED": False,
    "DATALOADER.NUM_WORKERS": 4, "DATALOADER.ASPECT_RATIO_GROUPING": True, "DATALOADER.FILTER_EMPTY_ANNOTATIONS": True,
    "TEST.DETECTIONS_PER_IMAGE": 100, "VIS_PERIOD": 0, "SEED": -1,
}


def test_base_tree_holds_the_upstream_defaults_of_survey_appendix_b():
    from omni3d_amd.d2.config import get_cfg
    a = _flat(get_cfg())
    wrong = {k: (a.get(k, "<missing>"), v) for k, v in UPSTREAM_DEFAULTS.items() if k not in a or a[k] != v}
    assert not wrong, wrong
