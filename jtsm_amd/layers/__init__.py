"""Operator layer of the MI355X JTSM hot path: the names model code imports from
``detectron2.layers`` / ``wsl.layers`` (detectron2/layers/__init__.py:1-13,
projects/WSL/wsl/layers/__init__.py:1-11) that lie on the path."""
from .moi_pool import MOIPool, moi_pool
from .roi_align import ROIAlign, roi_align
from .roi_align_rotated import ROIAlignRotated, roi_align_rotated

__all__ = [k for k in globals().keys() if not k.startswith("_")]
