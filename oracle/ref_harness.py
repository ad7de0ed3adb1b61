"""oracle/ref_harness.py -- TEST INFRASTRUCTURE.  Runs the reference's OWN hot-path files on CPU.

Only usable where /root/reference exists (the build container).  It installs the restated
upstream pieces of oracle/upstream.py under the `detectron2.*`, `pytorch3d.*`, `fvcore.*`
import paths (plus inert auto-stubs for everything cosmetic: cv2, renderer, pycocotools, ...),
then imports /root/reference/cubercnn unchanged and builds `RCNN3D` through the reference's own
`build_model` + YAML configs.  `oracle/make_golden.py` uses it to write tests/golden/*.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get("OMNI3D_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True   # the reference tree is read-only: never leave __pycache__ behind in it
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_STUB_ROOTS = ("detectron2", "pytorch3d", "fvcore", "cv2", "pycocotools", "termcolor", "torchvision", "matplotlib",
               "iopath")


class _DummyMeta(type):
    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _make_dummy(name)


def _make_dummy(name):
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return _make_dummy(item)

    return _DummyMeta(name, (), {"__init__": __init__, "__call__": __call__, "__getattr__": __getattr__})


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        d = _make_dummy(name)
        setattr(self, name, d)
        return d


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def _mod(name, **attrs):
    parts = name.split(".")
    for i in range(1, len(parts) + 1):
        n = ".".join(parts[:i])
        if n not in sys.modules:
            m = _StubModule(n)
            m.__path__ = []
            sys.modules[n] = m
            if i > 1:
                setattr(sys.modules[".".join(parts[:i - 1])], parts[i - 1], m)
    m = sys.modules[name]
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


_installed = False


def install():
    """Idempotent.  After this, `import cubercnn` resolves to /root/reference/cubercnn."""
    global _installed
    if _installed:
        return
    assert os.path.isdir(os.path.join(REFERENCE, "cubercnn")), f"reference checkout not found at {REFERENCE}"
    from oracle import upstream as U
    from omni3d_amd.d2 import comm

    ident = lambda f: f  # noqa: E731
    _mod("detectron2.config", get_cfg=U.get_cfg, CfgNode=U.CfgNode, configurable=U.configurable)
    _mod("detectron2.layers", ShapeSpec=U.ShapeSpec, cat=U.cat, nonzero_tuple=U.nonzero_tuple, batched_nms=U.batched_nms,
         cross_entropy=U.cross_entropy)
    _mod("detectron2.structures", Boxes=U.Boxes, Instances=U.Instances, ImageList=U.ImageList, BoxMode=U.BoxMode,
         pairwise_iou=U.pairwise_iou, pairwise_ioa=U.pairwise_ioa)
    _mod("detectron2.utils.events", get_event_storage=U.get_event_storage, EventStorage=U.EventStorage)
    _mod("detectron2.utils.registry", Registry=U.Registry)
    sys.modules["detectron2.utils.comm"] = comm
    _mod("detectron2.utils").comm = comm
    _mod("detectron2.utils.memory", retry_if_cuda_oom=ident)
    # dataset / evaluator plumbing of the reference (cubercnn/data/datasets.py, evaluation/omni3d_evaluation.py) as an oracle for
    # omni3d_amd/cubercnn/data/datasets.py: the catalogs are the plain registries of omni3d_amd.d2.data (no arithmetic), the
    # pycocotools / PathManager / Timer pieces are restated in oracle/upstream.py
    from omni3d_amd.d2.data import DatasetCatalog, MetadataCatalog
    _mod("detectron2.data", MetadataCatalog=MetadataCatalog, DatasetCatalog=DatasetCatalog)
    _mod("pycocotools.coco", COCO=U.COCO)
    _mod("pycocotools.mask", iou=U.coco_box_iou)
    _mod("detectron2.utils.file_io", PathManager=U.PathManagerLocal)
    _mod("fvcore.common.timer", Timer=U.Timer)
    _mod("detectron2.utils.logger", _log_api_usage=lambda *a, **k: None)
    _mod("detectron2.modeling", PROPOSAL_GENERATOR_REGISTRY=U.PROPOSAL_GENERATOR_REGISTRY)
    _mod("detectron2.modeling.backbone", Backbone=U.Backbone, BACKBONE_REGISTRY=U.BACKBONE_REGISTRY)
    _mod("detectron2.modeling.backbone.build", BACKBONE_REGISTRY=U.BACKBONE_REGISTRY)
    _mod("detectron2.modeling.backbone.fpn", FPN=U.FPN, LastLevelMaxPool=U.LastLevelMaxPool)
    _mod("detectron2.modeling.backbone.resnet", build_resnet_backbone=U.build_resnet_backbone)
    _mod("torchvision.models", resnet18=U.tv_resnet18, resnet34=U.tv_resnet34, resnet50=U.tv_resnet50, resnet101=U.tv_resnet101, densenet121=U.tv_densenet121,
         mnasnet1_0=U.tv_mnasnet1_0, shufflenet_v2_x1_0=U.tv_shufflenet_v2_x1_0)
    _mod("detectron2.modeling.proposal_generator", RPN=U.RPN, build_proposal_generator=U.build_proposal_generator)
    _mod("detectron2.modeling.proposal_generator.proposal_utils", add_ground_truth_to_proposals=U.add_ground_truth_to_proposals)
    _mod("detectron2.modeling.box_regression", Box2BoxTransform=U.Box2BoxTransform,
         _dense_box_regression_loss=U._dense_box_regression_loss)
    _mod("detectron2.modeling.roi_heads", StandardROIHeads=U.StandardROIHeads, ROI_HEADS_REGISTRY=U.ROI_HEADS_REGISTRY,
         select_foreground_proposals=U.select_foreground_proposals)
    _mod("detectron2.modeling.roi_heads.fast_rcnn", FastRCNNOutputLayers=U.FastRCNNOutputLayers,
         _log_classification_stats=U._log_classification_stats)
    _mod("detectron2.modeling.poolers", ROIPooler=U.ROIPooler)
    _mod("detectron2.modeling.meta_arch", META_ARCH_REGISTRY=U.META_ARCH_REGISTRY, GeneralizedRCNN=U.GeneralizedRCNN)
    _mod("detectron2.solver.build", maybe_add_gradient_clipping=lambda cfg, opt: opt)
    _mod("fvcore.nn", smooth_l1_loss=U.smooth_l1_loss)
    _mod("fvcore.nn.weight_init", c2_xavier_fill=U.c2_xavier_fill, c2_msra_fill=U.c2_msra_fill)
    _mod("pytorch3d", _C=U._C)
    _mod("pytorch3d.transforms", rotation_6d_to_matrix=U.rotation_6d_to_matrix, axis_angle_to_matrix=U.axis_angle_to_matrix,
         quaternion_to_matrix=U.quaternion_to_matrix, euler_angles_to_matrix=U.euler_angles_to_matrix)
    _mod("pytorch3d.transforms.rotation_conversions", _copysign=U._copysign)
    _mod("pytorch3d.transforms.so3", so3_relative_angle=U.so3_relative_angle)
    _mod("pytorch3d.ops.iou_box3d", _box_planes=U._box_planes, _box_triangles=U._box_triangles)
    sys.meta_path.append(_StubFinder())
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    _installed = True


def reference_cfg(config_name="cubercnn_DLA34_FPN.yaml", overrides=()):
    """cfg exactly as tools/train_net.py:318-349 builds it (get_cfg + get_cfg_defaults + YAML)."""
    install()
    from oracle import upstream as U
    from cubercnn.config import get_cfg_defaults
    cfg = U.get_cfg()
    get_cfg_defaults(cfg)
    cfg.merge_from_file(os.path.join(REFERENCE, "configs", config_name))
    base = ["MODEL.DEVICE", "cpu", "VIS_PERIOD", 0, "MODEL.WEIGHTS", "synthetic://random-init"]
    cfg.merge_from_list(base + list(overrides))
    return cfg


def build_reference_model(cfg, priors):
    install()
    # same registrations tools/train_net.py:40-50 triggers by importing these packages
    import cubercnn.modeling.backbone  # noqa: F401
    import cubercnn.modeling.proposal_generator  # noqa: F401
    import cubercnn.modeling.roi_heads  # noqa: F401
    from cubercnn.modeling.meta_arch import build_model
    return build_model(cfg, priors=priors)
