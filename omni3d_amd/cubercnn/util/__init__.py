from .math_util import *  # noqa: F401,F403
