from .dla import *  # noqa: F401,F403
from .fpn import FPN, Backbone  # noqa: F401
