"""detector_postprocess / sem_seg_postprocess — surface of detectron2/modeling/postprocessing.py:10-100.
Mask pasting and the bilinear resize are library calls (jtsm_amd/csrc/postprocess.hip)."""
import torch

from ..layers.mask_ops import paste_masks_in_image
from ..layers.postprocess import resize_bilinear
from ..structures import Instances


@torch.no_grad()
def detector_postprocess(results: Instances, output_height: int, output_width: int, mask_threshold: float = 0.5):
    """One image's detections at the requested output resolution: boxes rescaled and clipped, empty ones dropped,
    the (N, 1, M, M) mask probabilities pasted into (N, oh, ow) bitmaps."""
    in_h, in_w = results.image_size
    out = Instances((output_height, output_width), **results.get_fields())
    field = next((f for f in ("pred_boxes", "proposal_boxes") if out.has(f)), None)
    if field is None:
        raise AssertionError("Predictions must contain boxes!")
    boxes = out.get(field)
    boxes.scale(output_width / in_w, output_height / in_h)
    boxes.clip(out.image_size)
    out = out[boxes.nonempty()]
    if out.has("pred_masks"):
        out.pred_masks = paste_masks_in_image(out.pred_masks[:, 0], out.pred_boxes, out.image_size,
                                              threshold=mask_threshold)
    return out


@torch.no_grad()
def sem_seg_postprocess(result, img_size, output_height, output_width):
    """(C, H, W) logits: crop the padding away, bilinear resize to the output resolution -> (C, oh, ow)."""
    if tuple(result.shape[-2:]) == tuple(img_size) == (output_height, output_width):
        return result          # scale 1, no padding: the interpolation is the identity (weights 1 and 0)
    return resize_bilinear(result.unsqueeze(0), (output_height, output_width), crop_hw=img_size)[0]
