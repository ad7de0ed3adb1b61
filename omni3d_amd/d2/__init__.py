"""Host-side Detectron2-compatible surface (the drop-in boundary).

The reference (facebookresearch/omni3d) plugs its model into Detectron2 registries and reads a
yacs-style ``cfg``.  detectron2 / fvcore / yacs are not installable on the MI355X image, so this
package provides the *plumbing* the reference API needs -- Registry, CfgNode (+ ``_BASE_`` YAML
chains, upstream default keys), ``configurable``, ShapeSpec, Boxes / Instances / ImageList,
EventStorage, comm -- with the same names, argument meaning and error behaviour.  It contains no
arithmetic beyond trivial box bookkeeping; all hot-path math lives in omni3d_amd/csrc.
``omni3d_amd.install()`` aliases it under ``detectron2.*`` import paths when the real package
is absent so that tools/train_net.py's imports resolve.
"""
from .registry import Registry  # noqa: F401
from .config import CfgNode, get_cfg, configurable  # noqa: F401
from .layers import ShapeSpec, cat, nonzero_tuple  # noqa: F401
from .structures import Boxes, Instances, ImageList, BoxMode  # noqa: F401
from .events import EventStorage, get_event_storage  # noqa: F401
from . import comm  # noqa: F401
