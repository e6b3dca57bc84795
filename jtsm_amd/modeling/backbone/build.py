"""BACKBONE_REGISTRY / build_backbone — detectron2/modeling/backbone/build.py:7-33."""
from ...layers.shape_spec import ShapeSpec
from ...utils.registry import Registry
from .backbone import Backbone

BACKBONE_REGISTRY = Registry("BACKBONE")


def build_backbone(cfg, input_shape=None):
    if input_shape is None:
        input_shape = ShapeSpec(channels=len(cfg.MODEL.PIXEL_MEAN))
    backbone = BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, input_shape)
    assert isinstance(backbone, Backbone)
    return backbone
