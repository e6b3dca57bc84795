"""paste_masks_in_image — surface of detectron2/layers/mask_ops.py:74-145.  The reference's GPU branch samples every
mask over the whole image with grid_sample in chunks bounded by a 1 GB temporary; here one launch writes the (N, H, W)
byte result directly (csrc/postprocess.hip: paste_masks_kernel), no float temporary."""
import torch

from ..structures import Boxes
from .postprocess import paste_masks


@torch.no_grad()
def paste_masks_in_image(masks: torch.Tensor, boxes, image_shape, threshold: float = 0.5):
    assert masks.shape[-1] == masks.shape[-2], "Only square mask predictions are supported"
    N = len(masks)
    if N == 0:
        return masks.new_empty((0,) + tuple(image_shape), dtype=torch.uint8)
    if isinstance(boxes, Boxes):
        boxes = boxes.tensor
    assert len(boxes) == N, boxes.shape
    img_h, img_w = image_shape
    return paste_masks(masks, boxes, int(img_h), int(img_w), threshold)
