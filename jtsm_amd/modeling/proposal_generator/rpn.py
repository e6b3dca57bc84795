"""RPN — call surface of detectron2/modeling/proposal_generator/rpn.py:64-504 (StandardRPNHead, RPN) for BASELINE
configs[0] (configs/COCO-Detection/faster_rcnn_R_50_FPN_1x.yaml).  The shared 3x3 convolution (+ReLU) and the two 1x1
predictors are MFMA launches (layers/wrappers.py); anchor matching / sampling / proposal selection are device-side
torch ops and one batched-NMS library call per image.  Plumbing around the hot path, not part of it."""
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F
from torch import nn

from ...layers.shape_spec import ShapeSpec
from ...layers.wrappers import Conv2d, cat
from ...structures import Boxes, ImageList, Instances, pairwise_iou
from ...utils.registry import Registry
from ..anchor_generator import build_anchor_generator
from ..box_regression import Box2BoxTransform
from ..matcher import Matcher
from ..sampling import subsample_labels
from .build import PROPOSAL_GENERATOR_REGISTRY
from .proposal_utils import find_top_rpn_proposals

RPN_HEAD_REGISTRY = Registry("RPN_HEAD")


def build_rpn_head(cfg, input_shape):
    return RPN_HEAD_REGISTRY.get(cfg.MODEL.RPN.HEAD_NAME)(cfg, input_shape)


@RPN_HEAD_REGISTRY.register()
class StandardRPNHead(nn.Module):
    """conv3x3 + ReLU, then objectness (A) and anchor-delta (A * box_dim) 1x1 convolutions, shared by all levels."""

    def __init__(self, cfg, input_shape: List[ShapeSpec]):
        super().__init__()
        channels = {s.channels for s in input_shape}
        if len(channels) != 1:
            raise ValueError("StandardRPNHead: every level must have the same channel count")
        c = channels.pop()
        gen = build_anchor_generator(cfg, input_shape)
        per_cell = set(gen.num_anchors)
        if len(per_cell) != 1:
            raise ValueError("StandardRPNHead: every level must have the same number of anchors per position")
        a = per_cell.pop()
        self.box_dim = gen.box_dim
        self.conv = Conv2d(c, c, kernel_size=3, stride=1, padding=1, activation=F.relu)
        self.objectness_logits = Conv2d(c, a, kernel_size=1, stride=1)
        self.anchor_deltas = Conv2d(c, a * self.box_dim, kernel_size=1, stride=1)
        for layer in (self.conv, self.objectness_logits, self.anchor_deltas):
            nn.init.normal_(layer.weight, std=0.01)
            nn.init.constant_(layer.bias, 0)

    def forward(self, features: List[torch.Tensor]):
        logits, deltas = [], []
        for x in features:
            t = self.conv(x)
            logits.append(self.objectness_logits(t))
            deltas.append(self.anchor_deltas(t))
        return logits, deltas


@PROPOSAL_GENERATOR_REGISTRY.register()
class RPN(nn.Module):
    def __init__(self, cfg, input_shape: Dict[str, ShapeSpec]):
        super().__init__()
        r = cfg.MODEL.RPN
        self.in_features = r.IN_FEATURES
        shapes = [input_shape[f] for f in self.in_features]
        self.rpn_head = build_rpn_head(cfg, shapes)
        self.anchor_generator = build_anchor_generator(cfg, shapes)
        self.anchor_matcher = Matcher(r.IOU_THRESHOLDS, r.IOU_LABELS, allow_low_quality_matches=True)
        self.box2box_transform = Box2BoxTransform(weights=r.BBOX_REG_WEIGHTS)
        self.batch_size_per_image = r.BATCH_SIZE_PER_IMAGE
        self.positive_fraction = r.POSITIVE_FRACTION
        self.pre_nms_topk = {True: r.PRE_NMS_TOPK_TRAIN, False: r.PRE_NMS_TOPK_TEST}
        self.post_nms_topk = {True: r.POST_NMS_TOPK_TRAIN, False: r.POST_NMS_TOPK_TEST}
        self.nms_thresh = r.NMS_THRESH
        self.min_box_size = float(cfg.MODEL.PROPOSAL_GENERATOR.MIN_SIZE)
        self.anchor_boundary_thresh = r.BOUNDARY_THRESH
        self.loss_weight = {"loss_rpn_cls": r.LOSS_WEIGHT, "loss_rpn_loc": r.BBOX_REG_LOSS_WEIGHT * r.LOSS_WEIGHT}
        if r.BBOX_REG_LOSS_TYPE != "smooth_l1":
            raise NotImplementedError("RPN.BBOX_REG_LOSS_TYPE: smooth_l1 only")
        self.smooth_l1_beta = r.SMOOTH_L1_BETA

    # ------------------------------------------------------------------ labels
    @torch.no_grad()
    def label_and_sample_anchors(self, anchors: List[Boxes], gt_instances: List[Instances]):
        """Per image: labels (-1 ignore / 0 negative / 1 positive) for every anchor after sub-sampling to
        batch_size_per_image, and the matched ground-truth box of every anchor (rpn.py:283-335)."""
        anchors = Boxes.cat(anchors)
        labels, matched = [], []
        for inst in gt_instances:
            gt = inst.gt_boxes
            idx, lab = self.anchor_matcher(pairwise_iou(gt, anchors))
            lab = lab.to(device=gt.tensor.device)
            if self.anchor_boundary_thresh >= 0:
                raise NotImplementedError("RPN.BOUNDARY_THRESH >= 0 (legacy) is not supported")
            pos, neg = subsample_labels(lab, self.batch_size_per_image, self.positive_fraction, 0)
            sampled = torch.full_like(lab, -1)
            sampled[pos] = 1
            sampled[neg] = 0
            labels.append(sampled)
            matched.append(gt.tensor[idx] if len(gt) else torch.zeros_like(anchors.tensor))
        return labels, matched

    def losses(self, anchors, pred_objectness_logits, gt_labels, pred_anchor_deltas, gt_boxes):
        n = len(gt_labels)
        labels = torch.stack(gt_labels)                                  # (N, sum A)
        pos = labels == 1
        flat = Boxes.cat(anchors).tensor
        target = torch.stack([self.box2box_transform.get_deltas(flat, b) for b in gt_boxes])
        diff = (cat(pred_anchor_deltas, dim=1)[pos] - target[pos]).abs()
        if self.smooth_l1_beta >= 1e-5:
            b = self.smooth_l1_beta
            diff = torch.where(diff < b, 0.5 * diff * diff / b, diff - 0.5 * b)
        loc = diff.sum()
        valid = labels >= 0
        cls = F.binary_cross_entropy_with_logits(cat(pred_objectness_logits, dim=1)[valid],
                                                 labels[valid].to(torch.float32), reduction="sum")
        norm = self.batch_size_per_image * n
        out = {"loss_rpn_cls": cls / norm, "loss_rpn_loc": loc / norm}
        return {k: v * self.loss_weight.get(k, 1.0) for k, v in out.items()}

    # ------------------------------------------------------------------ forward
    def forward(self, images: ImageList, features: Dict[str, torch.Tensor],
                gt_instances: Optional[List[Instances]] = None):
        feats = [features[f] for f in self.in_features]
        anchors = self.anchor_generator(feats)
        logits, deltas = self.rpn_head(feats)
        # (N, A, H, W) -> (N, H*W*A); (N, A*4, H, W) -> (N, H*W*A, 4): views of the channels_last results
        logits = [z.permute(0, 2, 3, 1).flatten(1) for z in logits]
        bd = self.anchor_generator.box_dim
        deltas = [d.view(d.shape[0], -1, bd, d.shape[-2], d.shape[-1]).permute(0, 3, 4, 1, 2).flatten(1, -2)
                  for d in deltas]
        losses = {}
        if self.training:
            assert gt_instances is not None, "RPN requires gt_instances in training!"
            labels, boxes = self.label_and_sample_anchors(anchors, gt_instances)
            losses = self.losses(anchors, logits, labels, deltas, boxes)
        return self.predict_proposals(anchors, logits, deltas, images.image_sizes), losses

    @torch.no_grad()
    def predict_proposals(self, anchors, logits, deltas, image_sizes):
        decoded = []
        for a, d in zip(anchors, deltas):
            n = d.shape[0]
            flat = a.tensor.unsqueeze(0).expand(n, -1, -1).reshape(-1, 4)
            decoded.append(self.box2box_transform.apply_deltas(d.detach().reshape(-1, 4), flat).view(n, -1, 4))
        return find_top_rpn_proposals(decoded, [z.detach() for z in logits], image_sizes, self.nms_thresh,
                                      self.pre_nms_topk[self.training], self.post_nms_topk[self.training],
                                      self.min_box_size, self.training)
