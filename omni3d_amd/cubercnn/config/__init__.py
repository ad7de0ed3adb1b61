from .config import get_cfg_defaults  # noqa: F401
