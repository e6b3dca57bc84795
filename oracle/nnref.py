"""ORACLE — TEST INFRASTRUCTURE ONLY.

torch-CPU restatement of the dense layers on the hot path, written with the same stock torch
operators the reference composes ("Restate on torch-CPU"):
  conv_bn_act    detectron2/layers/wrappers.py:62-83 (Conv2d.forward: F.conv2d -> norm -> activation)
                 + layers/batch_norm.py:45-66 (FrozenBatchNorm2d: x*scale + bias)
                 + modeling/backbone/resnet.py:195-211 (out += shortcut; relu)
  linear         torch.nn.Linear as used at projects/WSL/wsl/modeling/roi_heads/box_head.py:90-93
All inputs/outputs are plain NCHW float32/float64 CPU tensors.
"""
import torch
import torch.nn.functional as F


def conv_bn_act(x, w, stride=1, pad=0, dil=1, scale=None, bias=None, residual=None, relu=False):
    y = F.conv2d(x, w, None, stride, pad, dil)
    if scale is not None:
        y = y * scale.view(1, -1, 1, 1)
    if bias is not None:
        y = y + bias.view(1, -1, 1, 1)
    if residual is not None:
        y = y + residual
    if relu:
        y = F.relu(y)
    return y


def linear(x, w, bias=None, relu=False):
    y = F.linear(x, w, bias)
    return F.relu(y) if relu else y
