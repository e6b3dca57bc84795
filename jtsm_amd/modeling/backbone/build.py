"""BACKBONE_REGISTRY and build_backbone(cfg[, input_shape]) — detectron2/modeling/backbone/build.py:7-33.
Registered builders take (cfg, ShapeSpec of the input image) and return a Backbone."""
from ...layers.shape_spec import ShapeSpec
from ...utils.registry import Registry
from .backbone import Backbone

BACKBONE_REGISTRY = Registry("BACKBONE")


def build_backbone(cfg, input_shape=None):
    shape = input_shape if input_shape is not None else ShapeSpec(channels=len(cfg.MODEL.PIXEL_MEAN))
    builder = BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)
    net = builder(cfg, shape)
    if not isinstance(net, Backbone):
        raise TypeError("%s returned %s, expected a Backbone" % (cfg.MODEL.BACKBONE.NAME, type(net).__name__))
    return net
