from .rcnn3d import RCNN3D, build_backbone, build_model  # noqa: F401
