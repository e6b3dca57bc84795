"""DiscriminativeAdaptionNeck — surface of projects/WSL/wsl/modeling/roi_heads/box_head.py:18-106.

flatten -> [Linear -> ReLU -> Dropout(0.5)] x len(fc_dims).  Each Linear+ReLU is one MFMA GEMM
launch.  The pooled features arrive channels_last, i.e. flattened in (h, w, c) order, while the
reference's fc1 weight columns are in (c, h, w) order; the parameter keeps the REFERENCE order
(state_dict compatible) and is re-ordered on the fly (a 100 MB permute next to a 200 GFLOP GEMM).
"""
import numpy as np
import torch
from torch import nn

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
        for layer in self.fcs:
            torch.nn.init.normal_(layer.weight, std=0.005)
            torch.nn.init.constant_(layer.bias, 0.1)

    def forward(self, x):
        if x.dim() == 4:
            c, h, w = x.shape[1:]
            first = self.fcs[0]
            if x.is_contiguous(memory_format=torch.channels_last) and h * w > 1:
                x = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)                      # (h,w,c) order, a view
                w1 = first.weight.view(-1, c, h, w).permute(0, 2, 3, 1).reshape(first.weight.shape[0], -1)
            else:
                x, w1 = x.flatten(1), first.weight
        else:
            w1 = self.fcs[0].weight
        for k, fc in enumerate(self.fcs):
            weight = w1 if k == 0 else fc.weight
            from ...layers.conv import linear_fused
            x = linear_fused(x, weight, fc.bias, True, True)
            if self.training and self.dropout_p > 0:
                x = torch.nn.functional.dropout(x, self.dropout_p, True)
        return x

    @property
    def output_shape(self):
        o = self._output_size
        return ShapeSpec(channels=o) if isinstance(o, int) else ShapeSpec(channels=o[0], height=o[1], width=o[2])
