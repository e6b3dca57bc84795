"""TSMOutputLayers — surface of projects/WSL/wsl/modeling/roi_heads/fast_rcnn_tsm.py:450-598 (layers, forward),
:672-694 (losses), :790-854 (predict_boxes / predict_probs / predict_probs_img).  Two Linear(input -> K+S-1) layers
`cls`, `det`; scores = softmax_c(cls) * per-image softmax over proposals(det); loss = BCE of the clamped per-image
score sums.

Two ways in: the reference-shaped `forward(x, proposals) -> (scores, proposal_deltas)` / `losses(predictions,
proposals, gt_classes_img_oh)` / `predict_*` (what the other WSL heads of the reference call; scores come from the
same HIP kernel and carry autograd), and the fused `logits` + `score_and_loss` used by JTSMROIHeads, where forward,
loss and backward are three launches of jtsm_amd/csrc/wsl_losses.hip and the two Linears ride in the one predictor
GEMM."""
from typing import List, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from ...layers.wrappers import Linear, cat
from ...layers.wsl_losses import mil_loss, mil_scores
from ...structures import Instances
from ..box_regression import Box2BoxTransform


class TSMOutputLayers(nn.Module):
    def __init__(self, input_size, *, num_classes, num_classes_stuff, box2box_transform=None, mean_loss=True,
                 loss_weight=1.0):
        super().__init__()
        self.num_classes = num_classes
        self.num_mil = num_classes + num_classes_stuff - 1
        self.box_dim = 4
        self.num_bbox_reg_classes = self.num_mil
        self.cls = Linear(input_size, self.num_mil)
        self.det = Linear(input_size, self.num_mil)
        nn.init.xavier_uniform_(self.cls.weight)
        nn.init.xavier_uniform_(self.det.weight)
        for l in [self.cls, self.det]:
            nn.init.constant_(l.bias, 0)
        self.box2box_transform = box2box_transform
        self.mean_loss = mean_loss
        self.loss_weight = {"loss_cls": loss_weight} if isinstance(loss_weight, float) else loss_weight

    @classmethod
    def from_config(cls, cfg, input_size):
        return cls(input_size, num_classes=cfg.MODEL.ROI_HEADS.NUM_CLASSES,
                   num_classes_stuff=cfg.MODEL.SEM_SEG_HEAD.NUM_CLASSES,
                   box2box_transform=Box2BoxTransform(weights=cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS),
                   mean_loss=cfg.WSL.MEAN_LOSS)

    def logits(self, x):
        if x.dim() > 2:
            x = torch.flatten(x, start_dim=1)
        return self.cls(x), self.det(x)

    def score_and_loss(self, cls_logits, det_logits, bag_offsets, gt_classes_img_oh, max_bag_rows):
        """-> (losses dict, scores (R,nc) detached, per-image probabilities (B,nc) detached)."""
        loss, scores, probs = mil_loss(cls_logits, det_logits, bag_offsets, gt_classes_img_oh, self.mean_loss,
                                       max_bag_rows)
        return {"loss_cls": loss * self.loss_weight.get("loss_cls", 1.0)}, scores, probs

    # ------------------------------------------------------------------ reference-shaped surface
    def forward(self, x, proposals: List[Instances] = None, context: bool = False):
        """-> (scores (R, K+S-1), proposal_deltas (R, 4 (K+S-1)) zeros) — fast_rcnn_tsm.py:548-598.  `proposals`
        gives the per-image bags (None / one image: a single bag)."""
        if context:
            raise NotImplementedError("the ContextLocNet variant (forward_contextlocnet) is outside the JTSM path")
        c, d = self.logits(x)
        counts = [len(p) for p in proposals] if proposals else [c.shape[0]]
        scores = mil_scores(c, d, counts)
        deltas = torch.zeros(scores.shape[0], self.num_bbox_reg_classes * self.box_dim, dtype=scores.dtype,
                             device=scores.device)
        return scores, deltas

    def predict_probs_img(self, predictions, proposals: List[Instances]):
        """Per-image class probabilities: sum of the bag's scores, clamped to [1e-6, 1 - 1e-6] (:840-854)."""
        scores, _ = predictions
        counts = [len(p) for p in proposals] if proposals else [scores.shape[0]]
        img = cat([s.sum(dim=0, keepdim=True) for s in scores.split(counts, dim=0)], dim=0)
        return torch.clamp(img, min=1e-6, max=1.0 - 1e-6)

    def losses(self, predictions, proposals: List[Instances], gt_classes_img_oh):
        """{"loss_cls": BCE(predict_probs_img, gt_classes_img_oh)} — mean, or sum / images (:672-694, :346-362)."""
        probs = self.predict_probs_img(predictions, proposals)
        target = gt_classes_img_oh.to(probs.dtype)
        if self.mean_loss:
            loss = F.binary_cross_entropy(probs, target, reduction="mean")
        else:
            loss = F.binary_cross_entropy(probs, target, reduction="sum") / target.size(0)
        return {"loss_cls": loss * self.loss_weight.get("loss_cls", 1.0)}

    def predict_probs(self, predictions: Tuple[torch.Tensor, torch.Tensor], proposals: List[Instances]):
        """Per image (R_i, K+S) : the scores with a zero background column appended (:816-838)."""
        scores, _ = predictions
        probs = torch.cat((scores, scores.new_zeros(scores.shape[0], 1)), dim=1)
        return probs.split([len(p) for p in proposals], dim=0)

    def predict_boxes(self, predictions: Tuple[torch.Tensor, torch.Tensor], proposals: List[Instances]):
        """Per image (R_i, 4 (K+S-1)): the (all-zero) deltas applied to the proposal boxes (:790-814)."""
        if not len(proposals):
            return []
        _, deltas = predictions
        boxes = cat([p.proposal_boxes.tensor for p in proposals], dim=0)
        return self.box2box_transform.apply_deltas(deltas, boxes).split([len(p) for p in proposals])
