"""detector_postprocess / sem_seg_postprocess — surface of detectron2/modeling/postprocessing.py:10-100.
Mask pasting and the bilinear resize are library calls (jtsm_amd/csrc/postprocess.hip)."""
import torch

from ..layers.mask_ops import paste_masks_in_image
from ..layers.postprocess import resize_bilinear
from ..structures import Instances


@torch.no_grad()
def detector_postprocess(results: Instances, output_height: int, output_width: int, mask_threshold: float = 0.5):
    """Rescale boxes to the output resolution, clip, drop empty ones, paste the (N,1,M,M) masks."""
    scale_x, scale_y = output_width / results.image_size[1], output_height / results.image_size[0]
    results = Instances((output_height, output_width), **results.get_fields())
    if results.has("pred_boxes"):
        output_boxes = results.pred_boxes
    elif results.has("proposal_boxes"):
        output_boxes = results.proposal_boxes
    else:
        raise AssertionError("Predictions must contain boxes!")
    output_boxes.scale(scale_x, scale_y)
    output_boxes.clip(results.image_size)
    results = results[output_boxes.nonempty()]
    if results.has("pred_masks"):
        results.pred_masks = paste_masks_in_image(results.pred_masks[:, 0, :, :], results.pred_boxes,
                                                  results.image_size, threshold=mask_threshold)
    return results


@torch.no_grad()
def sem_seg_postprocess(result, img_size, output_height, output_width):
    """(C, H, W) logits: crop the padding away, bilinear resize to the output resolution -> (C, oh, ow)."""
    if tuple(result.shape[-2:]) == tuple(img_size) == (output_height, output_width):
        return result          # scale 1, no padding: the interpolation is the identity (weights 1 and 0)
    return resize_bilinear(result.unsqueeze(0), (output_height, output_width), crop_hw=img_size)[0]
