"""pairwise IoU/IoA entry used by omni3d_amd.d2.structures (API parity with detectron2.structures)."""
from .det import pairwise_iou  # noqa: F401
