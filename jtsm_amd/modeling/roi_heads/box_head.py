"""DiscriminativeAdaptionNeck — surface of projects/WSL/wsl/modeling/roi_heads/box_head.py:18-106.

flatten -> [Linear -> ReLU -> Dropout(0.5)] x len(fc_dims).  Each Linear+ReLU is one MFMA GEMM
launch.  The pooled features arrive channels_last, i.e. flattened in (h, w, c) order, while the
reference's fc1 weight columns are in (c, h, w) order: the PARAMETER is stored in (h, w, c) column order
(no 100 MB permute on the step, planes cached like any other weight) and state_dict() / load_state_dict()
convert to and from the reference order, so checkpoints are interchangeable.
"""
import numpy as np
import torch
from torch import nn

from ...layers.fused_blocks import fc_stack_fused, fc_stack_ok

_FC_TAIL = __import__("os").environ.get("JTSM_FC_TAIL", "1") != "0"   # (A/B switch: the predictors' GEMM inside the stack's node)
from ...layers.shape_spec import ShapeSpec
from ...layers.wrappers import Conv2d, Linear
from ...utils.registry import Registry

ROI_BOX_HEAD_REGISTRY = Registry("ROI_BOX_HEAD")


def build_box_head(cfg, input_shape):
    return ROI_BOX_HEAD_REGISTRY.get(cfg.MODEL.ROI_BOX_HEAD.NAME)(cfg, input_shape)


@ROI_BOX_HEAD_REGISTRY.register()
class DiscriminativeAdaptionNeck(nn.Module):
    def __init__(self, cfg_or_shape, input_shape=None, *, conv_dims=None, fc_dims=None, conv_norm=""):
        super().__init__()
        if input_shape is not None:  # registry call: (cfg, input_shape)
            cfg = cfg_or_shape
            conv_dims = [cfg.MODEL.ROI_BOX_HEAD.CONV_DIM] * cfg.MODEL.ROI_BOX_HEAD.NUM_CONV
            fc_dims = cfg.MODEL.ROI_BOX_HEAD.DAN_DIM
            conv_norm = cfg.MODEL.ROI_BOX_HEAD.NORM
        else:
            input_shape = cfg_or_shape
        conv_dims, fc_dims = list(conv_dims or []), list(fc_dims or [])
        assert len(conv_dims) + len(fc_dims) > 0
        if conv_dims:
            raise NotImplementedError("DAN with conv layers (NUM_CONV > 0) is not used by any JTSM config")
        self._in_shape = (input_shape.channels, input_shape.height, input_shape.width)
        self._output_size = self._in_shape
        self.fcs = []
        for k, fc_dim in enumerate(fc_dims):
            fc = Linear(int(np.prod(self._output_size)), fc_dim)
            self.add_module("fc{}".format(k + 1), fc)
            self.fcs.append(fc)
            self._output_size = fc_dim
        self.dropout_p = 0.5
        self.takes_roi_scale = True      # forward(x, roi_scale=...): see JTSMROIHeads._box_features
        self._register_load_state_dict_pre_hook(self._to_hwc_hook)
        self._register_state_dict_hook(self._to_chw_hook)
        for layer in self.fcs:
            torch.nn.init.normal_(layer.weight, std=0.005)
            torch.nn.init.constant_(layer.bias, 0.1)

    # fc1.weight is STORED with its input columns in (h, w, c) order — the order a channels_last pooled feature
    # has in memory — so the 100 MB matrix is never re-ordered on the step.  Checkpoints keep the reference's
    # (c, h, w) column order: the two hooks below convert on load / save.
    def _hwc_cols(self, w, to_hwc):
        c, h, ww = self._in_shape
        if h * ww == 1:
            return w
        if to_hwc:
            return w.reshape(-1, c, h, ww).permute(0, 2, 3, 1).reshape(w.shape[0], -1)
        return w.reshape(-1, h, ww, c).permute(0, 3, 1, 2).reshape(w.shape[0], -1)

    def _to_hwc_hook(self, state_dict, prefix, *args):
        key = prefix + "fc1.weight"
        if key in state_dict:
            state_dict[key] = self._hwc_cols(state_dict[key], True)

    @staticmethod
    def _to_chw_hook(module, state_dict, prefix, local_metadata):
        key = prefix + "fc1.weight"
        if key in state_dict:
            state_dict[key] = module._hwc_cols(state_dict[key], False)

    def forward(self, x, roi_scale=None, tail=None):
        """roi_scale (R,), optional: the per-roi factor the features are multiplied by first
        (roi_heads_jtsm.py:607-633) — folded into the fused stack's plane split and data-gradient epilogue.
        tail = (weights, biases) of the linear layers that read the features (the box predictors), optional: run as
        one more GEMM of the same autograd node when the fused stack takes the call — the return value is then
        (features, [one output per tail layer]); otherwise the features alone, and the caller applies its layers."""
        if x.dim() == 4:
            if x.shape[2] * x.shape[3] > 1:
                x = x.permute(0, 2, 3, 1)               # (h,w,c) order: a view of a channels_last tensor
            x = x.reshape(x.shape[0], -1)
        if fc_stack_ok(x, self.fcs):                    # one autograd node (layers/fused_blocks.py: _FcStackFn)
            if tail is not None and _FC_TAIL and all(w.shape[1] == self.fcs[-1].out_features for w in tail[0]):
                return fc_stack_fused(x, self.fcs, roi_scale, self.dropout_p if self.training else 0.0, tail=tail)
            return fc_stack_fused(x, self.fcs, roi_scale, self.dropout_p if self.training else 0.0)
        if roi_scale is not None:
            x = x * roi_scale.view(-1, 1)
        for k, fc in enumerate(self.fcs):
            x = fc(x, relu=True)                       # Linear + bias + ReLU: one MFMA GEMM launch
            if self.training and self.dropout_p > 0:
                x = torch.nn.functional.dropout(x, self.dropout_p, True)
        return x

    @property
    def output_shape(self):
        o = self._output_size
        return ShapeSpec(channels=o) if isinstance(o, int) else ShapeSpec(channels=o[0], height=o[1], width=o[2])


@ROI_BOX_HEAD_REGISTRY.register()
class FastRCNNConvFCHead(nn.Module):
    """[conv3x3 + ReLU] x NUM_CONV then [Linear + ReLU] x NUM_FC — detectron2/modeling/roi_heads/box_head.py:26-118
    (the standard box head of BASELINE configs[0]; flattening keeps the reference's (c, h, w) column order)."""

    def __init__(self, cfg, input_shape: ShapeSpec):
        super().__init__()
        b = cfg.MODEL.ROI_BOX_HEAD
        if b.NORM:
            raise NotImplementedError("FastRCNNConvFCHead: conv norm '%s' is not supported" % b.NORM)
        self._output_size = (input_shape.channels, input_shape.height, input_shape.width)
        self.conv_norm_relus, self.fcs = [], []
        for k in range(b.NUM_CONV):
            conv = Conv2d(self._output_size[0], b.CONV_DIM, kernel_size=3, padding=1, bias=True,
                          activation=torch.nn.functional.relu)
            nn.init.kaiming_normal_(conv.weight, mode="fan_out", nonlinearity="relu")
            nn.init.constant_(conv.bias, 0)
            self.add_module("conv{}".format(k + 1), conv)
            self.conv_norm_relus.append(conv)
            self._output_size = (b.CONV_DIM, self._output_size[1], self._output_size[2])
        for k in range(b.NUM_FC):
            fc = Linear(int(np.prod(self._output_size)), b.FC_DIM)
            nn.init.kaiming_uniform_(fc.weight, a=1)      # c2_xavier_fill
            nn.init.constant_(fc.bias, 0)
            self.add_module("fc{}".format(k + 1), fc)
            self.fcs.append(fc)
            self._output_size = b.FC_DIM
        if not (self.conv_norm_relus or self.fcs):
            raise ValueError("FastRCNNConvFCHead needs NUM_CONV + NUM_FC > 0")

    def forward(self, x):
        for layer in self.conv_norm_relus:
            x = layer(x)
        if self.fcs:
            if x.dim() > 2:
                x = x.reshape(x.shape[0], -1)             # logical (c, h, w) order, as torch.flatten gives
            for fc in self.fcs:
                x = fc(x, relu=True)
        return x

    @property
    def output_shape(self):
        o = self._output_size
        return ShapeSpec(channels=o) if isinstance(o, int) else ShapeSpec(channels=o[0], height=o[1], width=o[2])
