"""Feature pyramid behind the reference's names — API surface of detectron2/modeling/backbone/fpn.py (:16-160 FPN,
:173-185 LastLevelMaxPool, :209-229 build_resnet_fpn_backbone): same constructor, module names
(fpn_lateral{L}, fpn_output{L}) and output names (p{L}); the construction is this repo's own.

MI355X mapping: lateral 1x1 and output 3x3 convolutions are MFMA launches with the bias in the epilogue; the
top-down `nearest x2 + add` is one fused bandwidth kernel that also writes the merged map's operand planes for the
3x3 that follows.  The pyramid is a list of levels (log2 stride, bottom-up feature, channels); forward walks it from
the coarsest level down.
"""
import math

from torch import nn

from ...layers.batch_norm import get_norm
from ...layers.elementwise import subsample2, upsample2_add
from ...layers.conv import set_segment
from ...layers.grad_fan import fan_out
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


class FPN(Backbone):
    def __init__(self, bottom_up, in_features, out_channels, norm="", top_block=None, fuse_type="sum"):
        super().__init__()
        if not isinstance(bottom_up, Backbone):
            raise TypeError("FPN: bottom_up must be a Backbone")
        if not in_features:
            raise ValueError("FPN: in_features is empty")
        if fuse_type not in ("sum", "avg"):
            raise ValueError("FPN: fuse_type must be 'sum' or 'avg', got %r" % (fuse_type,))
        shapes = bottom_up.output_shape()
        self.levels = []   # finest first: (L = log2 stride, bottom-up feature name)
        for name in in_features:
            stride = shapes[name].stride
            level = int(math.log2(stride))
            if self.levels and level != self.levels[-1][0] + 1:
                raise ValueError("FPN: strides of %s must double from one feature to the next" % (list(in_features),))
            self.levels.append((level, name))
            bias = norm == ""
            lateral = Conv2d(shapes[name].channels, out_channels, kernel_size=1, bias=bias,
                             norm=get_norm(norm, out_channels))
            output = Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=bias,
                            norm=get_norm(norm, out_channels))
            # the output conv's input gradient is (through the top-down add) the lateral conv's dy: let the
            # data-gradient epilogue write its operand planes for that contraction
            output.emit_dx_planes = True
            for conv, kind in ((lateral, "lateral"), (output, "output")):
                _xavier(conv)
                self.add_module("fpn_%s%d" % (kind, level), conv)
        self.bottom_up, self.in_features, self.top_block = bottom_up, in_features, top_block
        self._fuse_type = fuse_type
        names = ["p%d" % lv for lv, _ in self.levels]
        strides = {"p%d" % lv: 2 ** lv for lv, _ in self.levels}
        if top_block is not None:
            top = self.levels[-1][0]
            for extra in range(1, top_block.num_levels + 1):
                names.append("p%d" % (top + extra))
                strides[names[-1]] = 2 ** (top + extra)
        self._declare_outputs(names, {n: out_channels for n in names}, strides)
        self._size_divisibility = 2 ** self.levels[-1][0]

    @property
    def size_divisibility(self):
        return self._size_divisibility

    def forward(self, x):
        set_segment("backbone")        # (measurement tag of the contraction launches, layers/conv.py)
        feats = self.bottom_up(x)
        set_segment("fpn")
        out, merged = {}, None
        for level, name in reversed(self.levels):            # coarsest level first
            lateral = getattr(self, "fpn_lateral%d" % level)(feats[name])
            if merged is None:
                merged = lateral
            else:
                merged = upsample2_add(merged, lateral)      # nearest x2 + add, one pass
                if self._fuse_type == "avg":
                    merged = merged / 2
            if level != self.levels[0][0]:
                # read twice — by this level's output convolution and by the next finer level's merge: one view each, so
                # that the two gradient terms meet in one map (layers/grad_fan.py)
                mine, merged = fan_out(merged, 2)
            else:
                mine = merged
            out["p%d" % level] = getattr(self, "fpn_output%d" % level)(mine)
        if self.top_block is not None:
            src = self.top_block.in_feature
            extra = self.top_block(feats[src] if src in feats else out[src])
            for i, t in enumerate(extra, start=1):
                out["p%d" % (self.levels[-1][0] + i)] = t
        return {n: out[n] for n in self._out_features}

    def output_shape(self):
        return {n: ShapeSpec(channels=self._out_feature_channels[n], stride=self._out_feature_strides[n])
                for n in self._out_features}


class LastLevelMaxPool(nn.Module):
    """P6 = max_pool2d(P5, kernel 1, stride 2) — a strided copy."""

    def __init__(self):
        super().__init__()
        self.num_levels = 1
        self.in_feature = "p5"

    def forward(self, x):
        return [subsample2(x)]


def fpn_from_cfg(cfg, bottom_up):
    f = cfg.MODEL.FPN
    return FPN(bottom_up=bottom_up, in_features=f.IN_FEATURES, out_channels=f.OUT_CHANNELS, norm=f.NORM,
               top_block=LastLevelMaxPool(), fuse_type=f.FUSE_TYPE)


@BACKBONE_REGISTRY.register()
def build_resnet_fpn_backbone(cfg, input_shape: ShapeSpec):
    return fpn_from_cfg(cfg, build_resnet_backbone(cfg, input_shape))


@BACKBONE_REGISTRY.register()
def build_wsl_resnet_v2_fpn_backbone(cfg, input_shape: ShapeSpec):
    """projects/WSL/wsl/modeling/backbone/resnet_wsl_v2.py:729-750."""
    from .resnet_wsl_v2 import build_wsl_resnet_v2_backbone
    return fpn_from_cfg(cfg, build_wsl_resnet_v2_backbone(cfg, input_shape))
