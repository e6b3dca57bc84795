"""FastRCNNOutputLayers — call surface of detectron2/modeling/roi_heads/fast_rcnn.py:44-600 for BASELINE configs[0]:
two Linear layers (MFMA GEMMs), softmax cross-entropy (mean over the sampled rois) and the box regression loss over
foreground rois (smooth-L1, summed, divided by the number of SAMPLED rois, fast_rcnn.py:253-299); inference through
the per-image detection routine of the WSL heads (score threshold, per-class NMS, top-k: csrc/postprocess.hip)."""
from typing import List, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from ...layers.wrappers import Linear, cat
from ...structures import Instances
from ..box_regression import Box2BoxTransform
from .fast_rcnn_oicr import fast_rcnn_inference


class FastRCNNOutputLayers(nn.Module):
    def __init__(self, cfg, input_shape):
        super().__init__()
        size = input_shape.channels * (input_shape.width or 1) * (input_shape.height or 1)
        b, h = cfg.MODEL.ROI_BOX_HEAD, cfg.MODEL.ROI_HEADS
        self.num_classes = h.NUM_CLASSES
        self.box2box_transform = Box2BoxTransform(weights=b.BBOX_REG_WEIGHTS)
        reg_classes = 1 if b.CLS_AGNOSTIC_BBOX_REG else self.num_classes
        self.cls_score = Linear(size, self.num_classes + 1)
        self.bbox_pred = Linear(size, reg_classes * 4)
        nn.init.normal_(self.cls_score.weight, std=0.01)
        nn.init.normal_(self.bbox_pred.weight, std=0.001)
        for layer in (self.cls_score, self.bbox_pred):
            nn.init.constant_(layer.bias, 0)
        if b.BBOX_REG_LOSS_TYPE != "smooth_l1":
            raise NotImplementedError("ROI_BOX_HEAD.BBOX_REG_LOSS_TYPE: smooth_l1 only")
        self.smooth_l1_beta = b.SMOOTH_L1_BETA
        self.loss_weight = {"loss_cls": 1.0, "loss_box_reg": b.BBOX_REG_LOSS_WEIGHT}
        self.test_score_thresh, self.test_nms_thresh = h.SCORE_THRESH_TEST, h.NMS_THRESH_TEST
        self.test_topk_per_image = cfg.TEST.DETECTIONS_PER_IMAGE

    def forward(self, x):
        if x.dim() > 2:
            x = x.reshape(x.shape[0], -1)
        return self.cls_score(x), self.bbox_pred(x)

    def losses(self, predictions, proposals: List[Instances]):
        scores, deltas = predictions
        gt_classes = cat([p.gt_classes for p in proposals], dim=0) if proposals else scores.new_empty(0, dtype=torch.long)
        if gt_classes.numel() == 0:
            return {"loss_cls": scores.sum() * 0.0, "loss_box_reg": deltas.sum() * 0.0}
        boxes = cat([p.proposal_boxes.tensor for p in proposals], dim=0)
        gt_boxes = cat([(p.gt_boxes if p.has("gt_boxes") else p.proposal_boxes).tensor for p in proposals], dim=0)
        loss_cls = F.cross_entropy(scores, gt_classes, reduction="mean")
        fg = torch.nonzero((gt_classes >= 0) & (gt_classes < self.num_classes)).squeeze(1)
        if deltas.shape[1] == 4:
            picked = deltas[fg]
        else:
            picked = deltas.view(-1, self.num_classes, 4)[fg, gt_classes[fg]]
        diff = (picked - self.box2box_transform.get_deltas(boxes[fg], gt_boxes[fg])).abs()
        if self.smooth_l1_beta >= 1e-5:
            b = self.smooth_l1_beta
            diff = torch.where(diff < b, 0.5 * diff * diff / b, diff - 0.5 * b)
        out = {"loss_cls": loss_cls, "loss_box_reg": diff.sum() / max(gt_classes.numel(), 1.0)}
        return {k: v * self.loss_weight.get(k, 1.0) for k, v in out.items()}

    @torch.no_grad()
    def inference(self, predictions: Tuple[torch.Tensor, torch.Tensor], proposals: List[Instances]):
        scores, deltas = predictions
        counts = [len(p) for p in proposals]
        boxes = self.box2box_transform.apply_deltas(deltas, cat([p.proposal_boxes.tensor for p in proposals], dim=0))
        probs = F.softmax(scores, dim=-1)
        results, kept, _, _ = fast_rcnn_inference(list(boxes.split(counts)), list(probs.split(counts)),
                                                  [p.image_size for p in proposals], self.test_score_thresh,
                                                  self.test_nms_thresh, self.test_topk_per_image)
        return results, kept
