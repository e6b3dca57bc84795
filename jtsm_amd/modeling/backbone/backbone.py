"""Backbone ABC — detectron2/modeling/backbone/backbone.py:10-53."""
from abc import ABCMeta, abstractmethod

from torch import nn

from ...layers.shape_spec import ShapeSpec


class Backbone(nn.Module, metaclass=ABCMeta):
    def __init__(self):
        super().__init__()

    @abstractmethod
    def forward(self):
        pass

    @property
    def size_divisibility(self) -> int:
        return 0

    def output_shape(self):
        return {
            name: ShapeSpec(channels=self._out_feature_channels[name], stride=self._out_feature_strides[name])
            for name in self._out_features
        }
