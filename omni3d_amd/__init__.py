"""omni3d_amd -- MI355X-native (gfx950) hot path of Cube R-CNN (facebookresearch/omni3d).

Hand-written HIP kernels behind a C ABI (include/omni3d_hip.h, omni3d_amd/csrc) plus the
host-side mirror of the reference's Detectron2 registry surface.  See DESIGN.md.
"""
__version__ = "0.1.0"
