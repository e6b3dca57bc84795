"""TSMOutputLayers — surface of projects/WSL/wsl/modeling/roi_heads/fast_rcnn_tsm.py:450-598 (layers),
:672-694 (losses), :840-854 (predict_probs_img).  Two Linear(input -> K+S-1) layers `cls`, `det`;
scores = softmax_c(cls) * per-image softmax over proposals(det); loss = BCE of the clamped per-image
score sums.  Forward + loss + backward are three launches of jtsm_amd/csrc/wsl_losses.hip."""
import torch
from torch import nn

from ...layers.wrappers import Linear
from ...layers.wsl_losses import mil_loss
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
