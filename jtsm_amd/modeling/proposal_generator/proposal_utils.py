"""Proposal selection — detectron2/modeling/proposal_generator/proposal_utils.py:12-170: per level keep the
pre_nms_topk best-scoring decoded anchors, clip, drop empty boxes, NMS per level (one batched_nms call per image on
the HIP device: csrc/postprocess.hip), keep the post_nms_topk best over all levels."""
import math
from typing import List, Tuple

import torch

from ...layers.nms import batched_nms
from ...structures import Boxes, Instances


@torch.no_grad()
def find_top_rpn_proposals(proposals: List[torch.Tensor], pred_objectness_logits: List[torch.Tensor],
                           image_sizes: List[Tuple[int, int]], nms_thresh: float, pre_nms_topk: int,
                           post_nms_topk: int, min_box_size: float, training: bool):
    """proposals[l]: (N, H_l*W_l*A, 4), logits[l]: (N, H_l*W_l*A) -> list of N Instances (proposal_boxes,
    objectness_logits), sorted by score."""
    boxes_l, scores_l, level_l = [], [], []
    for level, (boxes, logits) in enumerate(zip(proposals, pred_objectness_logits)):
        k = min(pre_nms_topk, logits.shape[1])
        top, idx = logits.sort(descending=True, dim=1)
        top, idx = top[:, :k], idx[:, :k]
        boxes_l.append(torch.gather(boxes, 1, idx.unsqueeze(2).expand(-1, -1, 4)))
        scores_l.append(top)
        level_l.append(torch.full((k,), level, dtype=torch.int64, device=logits.device))
    all_boxes, all_scores, all_levels = torch.cat(boxes_l, 1), torch.cat(scores_l, 1), torch.cat(level_l)
    out = []
    for n, size in enumerate(image_sizes):
        boxes, scores, levels = Boxes(all_boxes[n]), all_scores[n], all_levels
        finite = torch.isfinite(boxes.tensor).all(dim=1) & torch.isfinite(scores)
        if not bool(finite.all()):
            if training:
                raise FloatingPointError("Predicted boxes or scores contain Inf/NaN. Training has diverged.")
            boxes, scores, levels = boxes[finite], scores[finite], levels[finite]
        boxes.clip(size)
        keep = boxes.nonempty(threshold=min_box_size)
        if int(keep.sum()) != len(boxes):
            boxes, scores, levels = boxes[keep], scores[keep], levels[keep]
        keep = batched_nms(boxes.tensor, scores, levels, nms_thresh)[:post_nms_topk]
        out.append(Instances(size, proposal_boxes=boxes[keep], objectness_logits=scores[keep]))
    return out


def add_ground_truth_to_proposals(gt_boxes: List[Boxes], proposals: List[Instances]):
    """Append every image's ground-truth boxes to its proposals with an objectness logit of
    logit(1 - 1e-10) (proposal_utils.py:127-170)."""
    assert gt_boxes is not None and len(gt_boxes) == len(proposals)
    logit = math.log((1.0 - 1e-10) / (1 - (1.0 - 1e-10)))
    out = []
    for gt, prop in zip(gt_boxes, proposals):
        extra = Instances(prop.image_size, proposal_boxes=gt,
                          objectness_logits=torch.full((len(gt),), logit, device=prop.objectness_logits.device))
        out.append(Instances.cat([prop, extra]))
    return out
