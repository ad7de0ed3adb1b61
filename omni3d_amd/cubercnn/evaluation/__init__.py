from .omni3d_evaluation import box3d_overlap  # noqa: F401
