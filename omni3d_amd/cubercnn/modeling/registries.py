"""The Detectron2 registries the reference plugs into."""
from ...d2.registry import Registry

META_ARCH_REGISTRY = Registry("META_ARCH")
BACKBONE_REGISTRY = Registry("BACKBONE")
PROPOSAL_GENERATOR_REGISTRY = Registry("PROPOSAL_GENERATOR")
RPN_HEAD_REGISTRY = Registry("RPN_HEAD")
ANCHOR_GENERATOR_REGISTRY = Registry("ANCHOR_GENERATOR")
ROI_HEADS_REGISTRY = Registry("ROI_HEADS")
ROI_BOX_HEAD_REGISTRY = Registry("ROI_BOX_HEAD")
