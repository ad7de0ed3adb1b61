from .roi_heads import ROIHeads3D, build_roi_heads  # noqa: F401
from .cube_head import CubeHead, ROI_CUBE_HEAD_REGISTRY, build_cube_head  # noqa: F401
from .fast_rcnn import FastRCNNOutputs  # noqa: F401
