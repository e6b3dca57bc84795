"""FPN — surface of detectron2/modeling/backbone/fpn.py:16-160 (FPN), :173-185 (LastLevelMaxPool),
:209-229 (build_resnet_fpn_backbone).  Lateral / output convolutions are MFMA launches with the
bias in the epilogue; the top-down `upsample x2 (nearest) + add` is one fused bandwidth kernel."""
import math

import torch.nn.functional as F
from torch import nn

from ...layers.batch_norm import get_norm
from ...layers.elementwise import subsample2, upsample2_add
from ...layers.shape_spec import ShapeSpec
from ...layers.wrappers import Conv2d
from .backbone import Backbone
from .build import BACKBONE_REGISTRY
from .resnet import build_resnet_backbone


def _xavier(conv):
    """c2_xavier_fill (fvcore): kaiming_uniform(a=1), zero bias."""
    nn.init.kaiming_uniform_(conv.weight, a=1)
    if conv.bias is not None:
        nn.init.constant_(conv.bias, 0)


def _assert_strides_are_log2_contiguous(strides):
    for i, stride in enumerate(strides[1:], 1):
        assert stride == 2 * strides[i - 1], "Strides {} {} are not log2 contiguous".format(stride, strides[i - 1])


class FPN(Backbone):
    def __init__(self, bottom_up, in_features, out_channels, norm="", top_block=None, fuse_type="sum"):
        super().__init__()
        assert isinstance(bottom_up, Backbone)
        assert in_features, in_features
        input_shapes = bottom_up.output_shape()
        strides = [input_shapes[f].stride for f in in_features]
        in_channels_per_feature = [input_shapes[f].channels for f in in_features]
        _assert_strides_are_log2_contiguous(strides)
        lateral_convs, output_convs = [], []
        use_bias = norm == ""
        for idx, in_channels in enumerate(in_channels_per_feature):
            lateral_conv = Conv2d(in_channels, out_channels, kernel_size=1, bias=use_bias,
                                  norm=get_norm(norm, out_channels))
            output_conv = Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=use_bias,
                                 norm=get_norm(norm, out_channels))
            _xavier(lateral_conv)
            _xavier(output_conv)
            stage = int(math.log2(strides[idx]))
            self.add_module("fpn_lateral{}".format(stage), lateral_conv)
            self.add_module("fpn_output{}".format(stage), output_conv)
            # the output conv's input gradient is (through the top-down add) the lateral conv's dy: let the
            # data-gradient epilogue write its bf16 planes for that contraction
            output_conv.emit_dx_planes = True
            lateral_convs.append(lateral_conv)
            output_convs.append(output_conv)
        # top-down order: coarsest level first
        self.lateral_convs = lateral_convs[::-1]
        self.output_convs = output_convs[::-1]
        self.top_block = top_block
        self.in_features = in_features
        self.bottom_up = bottom_up
        self._out_feature_strides = {"p{}".format(int(math.log2(s))): s for s in strides}
        if self.top_block is not None:
            for s in range(stage, stage + self.top_block.num_levels):
                self._out_feature_strides["p{}".format(s + 1)] = 2 ** (s + 1)
        self._out_features = list(self._out_feature_strides.keys())
        self._out_feature_channels = {k: out_channels for k in self._out_features}
        self._size_divisibility = strides[-1]
        assert fuse_type in {"avg", "sum"}
        self._fuse_type = fuse_type

    @property
    def size_divisibility(self):
        return self._size_divisibility

    def forward(self, x):
        bottom_up_features = self.bottom_up(x)
        feats = [bottom_up_features[f] for f in self.in_features[::-1]]
        results = []
        prev_features = self.lateral_convs[0](feats[0])
        results.append(self.output_convs[0](prev_features))
        for features, lateral_conv, output_conv in zip(feats[1:], self.lateral_convs[1:], self.output_convs[1:]):
            lateral_features = lateral_conv(features)
            prev_features = upsample2_add(prev_features, lateral_features)   # nearest x2 + add, one pass
            if self._fuse_type == "avg":
                prev_features = prev_features / 2
            results.insert(0, output_conv(prev_features))
        if self.top_block is not None:
            if self.top_block.in_feature in bottom_up_features:
                top_block_in_feature = bottom_up_features[self.top_block.in_feature]
            else:
                top_block_in_feature = results[self._out_features.index(self.top_block.in_feature)]
            results.extend(self.top_block(top_block_in_feature))
        assert len(self._out_features) == len(results)
        return dict(zip(self._out_features, results))

    def output_shape(self):
        return {name: ShapeSpec(channels=self._out_feature_channels[name], stride=self._out_feature_strides[name])
                for name in self._out_features}


class LastLevelMaxPool(nn.Module):
    """P6 = max_pool2d(P5, kernel 1, stride 2) — a strided copy."""

    def __init__(self):
        super().__init__()
        self.num_levels = 1
        self.in_feature = "p5"

    def forward(self, x):
        return [subsample2(x)]


@BACKBONE_REGISTRY.register()
def build_resnet_fpn_backbone(cfg, input_shape: ShapeSpec):
    bottom_up = build_resnet_backbone(cfg, input_shape)
    return FPN(bottom_up=bottom_up, in_features=cfg.MODEL.FPN.IN_FEATURES, out_channels=cfg.MODEL.FPN.OUT_CHANNELS,
               norm=cfg.MODEL.FPN.NORM, top_block=LastLevelMaxPool(), fuse_type=cfg.MODEL.FPN.FUSE_TYPE)
