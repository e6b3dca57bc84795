"""DefaultAnchorGenerator — call surface of detectron2/modeling/anchor_generator.py:80-232: per feature map,
`len(sizes) * len(aspect_ratios)` cell anchors (area size^2, height / width = ratio, centred on 0) translated to every
grid position (x = (j + offset) * stride).  Anchors depend only on the map size, so they are cached per
(level, H, W) on the device instead of being rebuilt every step."""
import math
from typing import List

import torch
from torch import nn

from ..layers.shape_spec import ShapeSpec
from ..structures import Boxes
from ..utils.registry import Registry

ANCHOR_GENERATOR_REGISTRY = Registry("ANCHOR_GENERATOR")


def _per_level(params, n, name):
    """A flat list applies to every level; a list of one list too; otherwise one list per level."""
    if not isinstance(params[0], (list, tuple)):
        return [list(params)] * n
    if len(params) == 1:
        return [list(params[0])] * n
    if len(params) != n:
        raise ValueError("anchor generator: %d lists of %s for %d feature maps" % (len(params), name, n))
    return [list(p) for p in params]


@ANCHOR_GENERATOR_REGISTRY.register()
class DefaultAnchorGenerator(nn.Module):
    box_dim = 4

    def __init__(self, cfg=None, input_shape: List[ShapeSpec] = None, *, sizes=None, aspect_ratios=None, strides=None,
                 offset=0.5):
        super().__init__()
        if cfg is not None:
            a = cfg.MODEL.ANCHOR_GENERATOR
            sizes, aspect_ratios, offset = a.SIZES, a.ASPECT_RATIOS, a.OFFSET
            strides = [s.stride for s in input_shape]
        if not 0.0 <= offset < 1.0:
            raise ValueError("anchor offset must be in [0, 1)")
        self.strides, self.offset = list(strides), float(offset)
        n = len(self.strides)
        self._cells = [self.generate_cell_anchors(s, r)
                       for s, r in zip(_per_level(sizes, n, "sizes"), _per_level(aspect_ratios, n, "aspect ratios"))]
        self._grids = {}

    @property
    def num_anchors(self):
        return [c.shape[0] for c in self._cells]

    num_cell_anchors = num_anchors

    @staticmethod
    def generate_cell_anchors(sizes=(32, 64, 128, 256, 512), aspect_ratios=(0.5, 1, 2)):
        """(len(sizes) * len(ratios), 4) boxes centred on the origin: w = sqrt(area / ratio), h = ratio * w
        (anchor_generator.py:180-212), size-major order."""
        rows = []
        for size in sizes:
            area = float(size) ** 2
            for ratio in aspect_ratios:
                w = math.sqrt(area / ratio)
                h = ratio * w
                rows.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
        return torch.tensor(rows, dtype=torch.float32)

    def grid_anchors(self, level, h, w, device):
        key = (level, h, w, str(device))
        hit = self._grids.get(key)
        if hit is None:
            stride, cell = self.strides[level], self._cells[level].to(device)
            xs = torch.arange(self.offset * stride, w * stride, step=stride, dtype=torch.float32, device=device)
            ys = torch.arange(self.offset * stride, h * stride, step=stride, dtype=torch.float32, device=device)
            sy, sx = torch.meshgrid(ys, xs, indexing="ij")
            shifts = torch.stack((sx.reshape(-1), sy.reshape(-1), sx.reshape(-1), sy.reshape(-1)), dim=1)
            hit = self._grids[key] = (shifts.view(-1, 1, 4) + cell.view(1, -1, 4)).reshape(-1, 4)
        return hit

    def forward(self, features: List[torch.Tensor]):
        """-> list[Boxes], one per feature map, (H * W * A, 4) in (y, x, anchor) order."""
        return [Boxes(self.grid_anchors(i, f.shape[-2], f.shape[-1], f.device)) for i, f in enumerate(features)]


def build_anchor_generator(cfg, input_shape):
    return ANCHOR_GENERATOR_REGISTRY.get(cfg.MODEL.ANCHOR_GENERATOR.NAME)(cfg, input_shape)
