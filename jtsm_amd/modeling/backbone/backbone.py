"""Backbone base class — what model builders rely on from detectron2/modeling/backbone/backbone.py:10-53:
`forward(x) -> {name: feature}`, `output_shape() -> {name: ShapeSpec}`, `size_divisibility`.
Sub-classes describe their outputs once with `_declare_outputs`; the legacy attribute names
(`_out_features`, `_out_feature_channels`, `_out_feature_strides`) stay readable for code that pokes at them."""
from typing import Dict, Iterable

from torch import nn

from ...layers.shape_spec import ShapeSpec


class Backbone(nn.Module):
    def __init__(self):
        super().__init__()
        self._out_features = []
        self._out_feature_channels: Dict[str, int] = {}
        self._out_feature_strides: Dict[str, int] = {}

    def _declare_outputs(self, names: Iterable[str], channels: Dict[str, int], strides: Dict[str, int]):
        self._out_features = list(names)
        self._out_feature_channels = dict(channels)
        self._out_feature_strides = dict(strides)

    def forward(self, *args, **kwargs):
        raise NotImplementedError("%s must implement forward()" % type(self).__name__)

    @property
    def size_divisibility(self) -> int:
        """Inputs must be padded to a multiple of this (0 = no requirement)."""
        return 0

    def output_shape(self) -> Dict[str, ShapeSpec]:
        return {n: ShapeSpec(channels=self._out_feature_channels[n], stride=self._out_feature_strides[n])
                for n in self._out_features}
