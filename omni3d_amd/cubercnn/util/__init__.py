from .math_util import *  # noqa: F401,F403
from .math_util import R_from_allocentric, R_to_allocentric  # noqa: F401
from .util import CubeRCNNHandler, compute_priors, file_parts, load_json, save_json  # noqa: F401
