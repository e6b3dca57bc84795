"""SemSegFPNHead — surface of detectron2/modeling/meta_arch/semantic_seg.py:95-188.
Per FPN level: [conv3x3 -> GroupNorm(32) -> ReLU (-> bilinear x2)] x log2(stride/4), summed, 1x1
predictor; loss = CE(bilinear x4 of the logits, target, ignore 255) * LOSS_WEIGHT.
The 3x3 / 1x1 convolutions are MFMA launches, GroupNorm+ReLU and the x2 bilinear up-sampling are
channels-last HIP kernels, and the x4 up-sampling + cross-entropy is one fused kernel that never
materialises the full-resolution logits (all in csrc/semseg_ops.hip)."""
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from ...layers.shape_spec import ShapeSpec
from ...layers.wrappers import Conv2d
from ...utils.registry import Registry

SEM_SEG_HEADS_REGISTRY = Registry("SEM_SEG_HEADS")


class UpsampleBilinear2x(nn.Module):
    """nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False) on channels_last tensors."""

    def forward(self, x):
        from ...layers.elementwise import upsample_bilinear2x
        return upsample_bilinear2x(x)


def _upsample_logits(x, stride):
    """F.interpolate(x, scale_factor=stride, mode="bilinear", align_corners=False) (semantic_seg.py:172-176) as one
    library launch; result planar (N, C, H*stride, W*stride)."""
    from ...layers.postprocess import resize_bilinear
    s = int(stride)
    return resize_bilinear(x, (x.shape[2] * s, x.shape[3] * s), scale_factor=s)


def build_sem_seg_head(cfg, input_shape):
    return SEM_SEG_HEADS_REGISTRY.get(cfg.MODEL.SEM_SEG_HEAD.NAME)(cfg, input_shape)


def _scale_head(in_channels, width, stride, common_stride, norm):
    """One pyramid level's branch: log2(stride / common_stride) rounds (at least one) of conv3x3 -> norm -> ReLU, each
    followed by a x2 bilinear up-sampling while the level is coarser than the common stride.  The Sequential's
    positions (convolutions at 0, 2, 4, ... when up-sampling sits between them) are the checkpoint's key names."""
    rounds = max(1, int(round(np.log2(stride / common_stride))))
    ops = []
    for k in range(rounds):
        conv = Conv2d(in_channels if k == 0 else width, width, kernel_size=3, stride=1, padding=1, bias=not norm,
                      norm=nn.GroupNorm(32, width) if norm == "GN" else None, activation=F.relu)
        nn.init.kaiming_normal_(conv.weight, mode="fan_out", nonlinearity="relu")
        if conv.bias is not None:
            nn.init.zeros_(conv.bias)
        ops.append(conv)
        if stride != common_stride:
            ops.append(UpsampleBilinear2x())
    return nn.Sequential(*ops)


@SEM_SEG_HEADS_REGISTRY.register()
class SemSegFPNHead(nn.Module):
    def __init__(self, cfg, input_shape: Dict[str, ShapeSpec]):
        super().__init__()
        head = cfg.MODEL.SEM_SEG_HEAD
        if head.NORM not in ("", "GN"):
            raise NotImplementedError("SemSegFPNHead norm '%s' is not used on the JTSM path" % head.NORM)
        self.in_features = head.IN_FEATURES
        self.ignore_value, self.common_stride, self.loss_weight = head.IGNORE_VALUE, head.COMMON_STRIDE, head.LOSS_WEIGHT
        self.scale_heads = []
        for name in self.in_features:          # registered under the level's own name ("p2" .. "p5"), as in checkpoints
            branch = _scale_head(input_shape[name].channels, head.CONVS_DIM, input_shape[name].stride,
                                 self.common_stride, head.NORM)
            self.add_module(name, branch)
            self.scale_heads.append(branch)
        self.predictor = Conv2d(head.CONVS_DIM, head.NUM_CLASSES, kernel_size=1, stride=1, padding=0)
        nn.init.kaiming_normal_(self.predictor.weight, mode="fan_out", nonlinearity="relu")
        nn.init.zeros_(self.predictor.bias)

    def forward(self, features, targets=None):
        x = self.layers(features)
        if self.training:
            return None, self.losses(x, targets)
        return _upsample_logits(x, self.common_stride), {}

    def layers(self, features):
        from ...layers.elementwise import sum_tensors
        # `x = x + y` over the levels, in level order, as one pass that also writes the predictor's operand planes
        return self.predictor(sum_tensors([self.scale_heads[i](features[f]) for i, f in enumerate(self.in_features)]))

    def losses(self, predictions, targets):
        from ...layers.elementwise import semseg_cross_entropy
        if predictions.is_cuda and predictions.shape[1] <= 64 and self.common_stride == int(self.common_stride):
            loss = semseg_cross_entropy(predictions.float(), targets, int(self.common_stride), self.ignore_value)
        else:
            raise NotImplementedError("semantic loss: <= 64 classes on the HIP device only")
        return {"loss_sem_seg": loss * self.loss_weight}


@SEM_SEG_HEADS_REGISTRY.register()
class TwoClassHead(nn.Module):
    """The parameter-free semantic head of the shipped JTSM configs (projects/WSL/wsl/modeling/seg_heads/
    seg_heads.py:231-275): a constant two-class map ("everything is class 1"), no loss — those configs train the
    box and mask branches only."""

    def __init__(self, cfg, input_shape: Dict[str, ShapeSpec]):
        super().__init__()
        self.in_features = cfg.MODEL.SEM_SEG_HEAD.IN_FEATURES
        self.ignore_value = cfg.MODEL.SEM_SEG_HEAD.IGNORE_VALUE
        self.common_stride = cfg.MODEL.SEM_SEG_HEAD.COMMON_STRIDE
        self.loss_weight = cfg.MODEL.SEM_SEG_HEAD.LOSS_WEIGHT
        self.scale = int(input_shape[self.in_features[0]].stride / self.common_stride)

    def layers(self, features):
        f = features[self.in_features[0]]
        x = torch.zeros((f.size(0), 2, f.size(2) * self.scale, f.size(3) * self.scale), device=f.device,
                        dtype=torch.float32)
        x[:, 1] = 1
        return x

    def losses(self, predictions, targets):
        return {}

    def forward(self, features, targets=None):
        if self.training:
            return None, {}
        x = self.layers(features)
        return _upsample_logits(x, self.common_stride), {}
