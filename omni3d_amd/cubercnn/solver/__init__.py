from .build import FlatAdam, FlatSGD, build_optimizer, freeze_bn  # noqa: F401
from .checkpoint import PeriodicCheckpointerOnlyOne  # noqa: F401
from .guard import StepGuard, allreduce_dict  # noqa: F401
