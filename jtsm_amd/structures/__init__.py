"""Data ABI between model components (detectron2/structures): the subset the JTSM path touches."""
from .boxes import Boxes, pairwise_iou
from .image_list import ImageList
from .instances import Instances
from .rotated_boxes import RotatedBoxes

__all__ = ["Boxes", "pairwise_iou", "ImageList", "Instances", "RotatedBoxes"]
