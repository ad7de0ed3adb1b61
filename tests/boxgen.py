"""the IoU3D workload generator lives with the product's synthetic data (omni3d_amd/boxgen.py: bench.py and smoke() use it too);
tests import it under its old name"""
from omni3d_amd.boxgen import *  # noqa: F401,F403
from omni3d_amd.boxgen import UNIT, corners, omni3d_like_pairs, random_boxes  # noqa: F401
