"""CNNBlockBase — base of stem / residual blocks (detectron2/layers/blocks.py:16-55): remembers
(in_channels, out_channels, stride) and can freeze itself (parameters stop training, any BatchNorm
becomes FrozenBatchNorm2d)."""
from torch import nn

from .batch_norm import FrozenBatchNorm2d


class CNNBlockBase(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, stride: int):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride

    def freeze(self):
        self.requires_grad_(False)
        FrozenBatchNorm2d.convert_frozen_batchnorm(self)
        return self
