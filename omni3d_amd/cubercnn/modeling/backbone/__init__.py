from .dla import *  # noqa: F401,F403
from .resnet import ResNet, build_resnet_from_vision_fpn_backbone  # noqa: F401
from .densenet import DenseNetBackbone, build_densenet_fpn_backbone  # noqa: F401
from .mnasnet import MNASNetBackbone, build_mnasnet_fpn_backbone  # noqa: F401
from .shufflenet import ShufflenetBackbone, build_shufflenet_fpn_backbone  # noqa: F401
from .fpn import FPN, Backbone, LastLevelMaxPool  # noqa: F401
