"""OICROutputLayers — surface of projects/WSL/wsl/modeling/roi_heads/fast_rcnn_oicr.py:448-586 (layers,
losses) and :684-783 (predict_probs / predict_boxes).  `cls_score: Linear(in, K+1)`,
`bbox_pred: Linear(in, 4K)`; losses = instance-weighted CE and L1 (jtsm_amd/csrc/wsl_losses.hip)."""
import torch
import torch.nn.functional as F
from torch import nn

from ...layers.wrappers import Linear
from ...layers.wsl_losses import oicr_loss
from ..box_regression import Box2BoxTransform


class OICROutputLayers(nn.Module):
    def __init__(self, input_size, *, num_classes, box2box_transform, refine_k, refine_reg, loss_weight=1.0):
        super().__init__()
        self.num_classes = num_classes
        self.box_dim = len(box2box_transform.weights)
        self.num_bbox_reg_classes = num_classes
        self.cls_score = Linear(input_size, num_classes + 1)
        self.bbox_pred = Linear(input_size, num_classes * self.box_dim)
        nn.init.normal_(self.cls_score.weight, std=0.01)
        nn.init.normal_(self.bbox_pred.weight, std=0.001)
        for l in [self.cls_score, self.bbox_pred]:
            nn.init.constant_(l.bias, 0)
        self.box2box_transform = box2box_transform
        self.refine_k = refine_k
        self.refine_reg = refine_reg
        self.loss_weight = {"loss_box_reg": loss_weight} if isinstance(loss_weight, float) else loss_weight

    @classmethod
    def from_config(cls, cfg, input_size, refine_k):
        return cls(input_size, num_classes=cfg.MODEL.ROI_HEADS.NUM_CLASSES,
                   box2box_transform=Box2BoxTransform(weights=cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS),
                   refine_k=refine_k, refine_reg=cfg.WSL.REFINE_REG,
                   loss_weight={"loss_box_reg": cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_WEIGHT})

    @property
    def has_reg(self):
        return bool(self.refine_reg[self.refine_k])

    def forward(self, x):
        if x.dim() > 2:
            x = torch.flatten(x, start_dim=1)
        scores = self.cls_score(x)
        if self.has_reg:
            return scores, self.bbox_pred(x)
        return scores, torch.zeros(scores.shape[0], self.num_bbox_reg_classes * self.box_dim, dtype=scores.dtype,
                                   device=scores.device)

    def losses(self, predictions, proposal_boxes, gt_classes, gt_boxes, gt_weights):
        """predictions = (logits (R,K+1), deltas (R,4K)); the rest are cat'ed over images."""
        scores, deltas = predictions
        lc, lb = oicr_loss(scores, deltas if self.has_reg else None, gt_classes, gt_weights,
                           proposal_boxes if self.has_reg else None, gt_boxes if self.has_reg else None)
        k = "_r" + str(self.refine_k)
        out = {"loss_cls" + k: lc}
        if self.has_reg:
            out["loss_box_reg" + k] = lb * self.loss_weight.get("loss_box_reg", 1.0)
        return out

    def predict_probs(self, predictions, counts):
        return F.softmax(predictions[0], dim=-1).split(counts, dim=0)

    def predict_boxes(self, predictions, proposal_boxes, counts):
        return self.box2box_transform.apply_deltas(predictions[1], proposal_boxes).split(counts)
