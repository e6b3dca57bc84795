"""ROI heads base — the parts of projects/WSL/wsl/modeling/roi_heads/roi_heads.py on the JTSM path:
registry + build_roi_heads (:25-43), get_image_level_gt (:145-161), select_foreground_proposals,
ROIHeads.label_and_sample_proposals (:222-370; the WSL variant keeps EVERY proposal, :253-254)."""
from typing import List

import torch

from ...structures import Boxes, Instances, pairwise_iou
from ...utils.registry import Registry
from ..matcher import Matcher

ROI_HEADS_REGISTRY = Registry("ROI_HEADS")


def build_roi_heads(cfg, input_shape):
    return ROI_HEADS_REGISTRY.get(cfg.MODEL.ROI_HEADS.NAME)(cfg, input_shape)


def select_foreground_proposals(proposals: List[Instances], bg_label: int):
    fg_proposals, fg_selection_masks = [], []
    for proposals_per_image in proposals:
        gt_classes = proposals_per_image.gt_classes
        fg_selection_mask = (gt_classes != -1) & (gt_classes != bg_label)
        fg_idxs = fg_selection_mask.nonzero().squeeze(1)
        fg_proposals.append(proposals_per_image[fg_idxs])
        fg_selection_masks.append(fg_selection_mask)
    return fg_proposals, fg_selection_masks


@torch.no_grad()
def get_image_level_gt(targets, num_classes):
    """Per image: sorted unique thing classes (float/int64 copies) and their one-hot (B,num_classes)."""
    if targets is None:
        return None, None, None
    gt_classes_img = [torch.unique(t.gt_classes, sorted=True) for t in targets]
    gt_classes_img_int = [gt.to(torch.int64) for gt in gt_classes_img]
    oh = torch.zeros((len(targets), num_classes), dtype=torch.float, device=gt_classes_img[0].device)
    for i, gt in enumerate(gt_classes_img_int):
        oh[i, gt] = 1
    return gt_classes_img, gt_classes_img_int, oh


class ROIHeads(torch.nn.Module):
    def __init__(self, *, num_classes, batch_size_per_image, positive_fraction, proposal_matcher,
                 proposal_append_gt=True):
        super().__init__()
        self.batch_size_per_image = batch_size_per_image
        self.positive_fraction = positive_fraction
        self.num_classes = num_classes
        self.proposal_matcher = proposal_matcher
        self.proposal_append_gt = proposal_append_gt

    @classmethod
    def from_config(cls, cfg):
        return {
            "batch_size_per_image": cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE,
            "positive_fraction": cfg.MODEL.ROI_HEADS.POSITIVE_FRACTION,
            "num_classes": cfg.MODEL.ROI_HEADS.NUM_CLASSES,
            "proposal_append_gt": cfg.MODEL.ROI_HEADS.PROPOSAL_APPEND_GT,
            "proposal_matcher": Matcher(cfg.MODEL.ROI_HEADS.IOU_THRESHOLDS, cfg.MODEL.ROI_HEADS.IOU_LABELS,
                                        allow_low_quality_matches=False),
        }

    def _sample_proposals(self, matched_idxs, matched_labels, gt_classes):
        """No sub-sampling in the WSL heads: every proposal is kept and labelled."""
        if gt_classes.numel() > 0:
            gt_classes = gt_classes[matched_idxs]
            gt_classes[matched_labels == 0] = self.num_classes
            gt_classes[matched_labels == -1] = -1
        else:
            gt_classes = torch.zeros_like(matched_idxs) + self.num_classes
        return torch.arange(gt_classes.shape[0], device=gt_classes.device), gt_classes

    @torch.no_grad()
    def label_and_sample_proposals(self, proposals: List[Instances], targets: List[Instances], suffix=""):
        if self.proposal_append_gt:
            raise NotImplementedError("PROPOSAL_APPEND_GT must be False for the WSL heads "
                                      "(projects/WSL/configs/Base-RCNN-DilatedC5.yaml:15)")
        out = []
        for proposals_per_image, targets_per_image in zip(proposals, targets):
            has_gt = len(targets_per_image) > 0
            match_quality_matrix = pairwise_iou(targets_per_image.gt_boxes, proposals_per_image.proposal_boxes)
            matched_idxs, matched_labels = self.proposal_matcher(match_quality_matrix)
            sampled_idxs, gt_classes = self._sample_proposals(matched_idxs, matched_labels,
                                                              targets_per_image.gt_classes)
            proposals_per_image = proposals_per_image[sampled_idxs]
            proposals_per_image.gt_classes = gt_classes
            if has_gt:
                sampled_targets = matched_idxs[sampled_idxs]
                for name, value in targets_per_image.get_fields().items():
                    if name.startswith("gt_") and name != "gt_classes":
                        proposals_per_image.set(name, value[sampled_targets])
                proposals_per_image.set("matched_gt_idx", sampled_targets)
            out.append(proposals_per_image)
        return out
