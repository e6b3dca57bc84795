"""ShapeSpec(channels, height, width, stride) — the shape descriptor passed between model builders
(detectron2/layers/shape_spec.py).  All four fields are optional; instances are immutable and
unpack / compare like a 4-tuple."""
from dataclasses import astuple, dataclass
from typing import Optional


@dataclass(frozen=True)
class ShapeSpec:
    channels: Optional[int] = None
    height: Optional[int] = None
    width: Optional[int] = None
    stride: Optional[int] = None

    def __iter__(self):
        return iter(astuple(self))

    def _replace(self, **changes):
        data = dict(zip(("channels", "height", "width", "stride"), astuple(self)))
        data.update(changes)
        return ShapeSpec(**data)
