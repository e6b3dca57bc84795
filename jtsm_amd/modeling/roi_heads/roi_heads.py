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


@ROI_HEADS_REGISTRY.register()
class StandardROIHeads(torch.nn.Module):
    """The fully supervised box (+ mask) heads of BASELINE configs[0] — call surface of
    detectron2/modeling/roi_heads/roi_heads.py:520-875: ground truth appended to the proposals, IoU matching, 512
    sampled rois per image (25 % foreground), multi-level ROIAlignV2 -> box head -> FastRCNNOutputLayers; with
    MASK_ON, ROIAlign 14 x 14 of the foreground rois -> mask head -> per-class BCE against the matched instance's
    bitmask cropped to the roi (structures/masks.py:169-200, the same ROIAlign).  Plumbing over this repo's kernels."""

    def __init__(self, cfg, input_shape):
        super().__init__()
        from ...layers.shape_spec import ShapeSpec
        from ..poolers import ROIPooler
        from ..sampling import subsample_labels  # noqa: F401  (used in label_and_sample_proposals)
        from .box_head import build_box_head
        from .fast_rcnn import FastRCNNOutputLayers
        from .mask_head import build_mask_head

        h = cfg.MODEL.ROI_HEADS
        self.num_classes, self.batch_size_per_image = h.NUM_CLASSES, h.BATCH_SIZE_PER_IMAGE
        self.positive_fraction, self.proposal_append_gt = h.POSITIVE_FRACTION, h.PROPOSAL_APPEND_GT
        self.proposal_matcher = Matcher(h.IOU_THRESHOLDS, h.IOU_LABELS, allow_low_quality_matches=False)
        self.in_features = h.IN_FEATURES
        scales = tuple(1.0 / input_shape[k].stride for k in self.in_features)
        channels = {input_shape[f].channels for f in self.in_features}
        assert len(channels) == 1, channels
        c = channels.pop()
        b = cfg.MODEL.ROI_BOX_HEAD
        self.box_pooler = ROIPooler(output_size=b.POOLER_RESOLUTION, scales=scales, sampling_ratio=b.POOLER_SAMPLING_RATIO,
                                    pooler_type=b.POOLER_TYPE)
        self.box_head = build_box_head(cfg, ShapeSpec(channels=c, height=b.POOLER_RESOLUTION, width=b.POOLER_RESOLUTION))
        self.box_predictor = FastRCNNOutputLayers(cfg, self.box_head.output_shape)
        self.mask_on = cfg.MODEL.MASK_ON
        if self.mask_on:
            m = cfg.MODEL.ROI_MASK_HEAD
            self.mask_pooler = ROIPooler(output_size=m.POOLER_RESOLUTION, scales=scales,
                                         sampling_ratio=m.POOLER_SAMPLING_RATIO, pooler_type=m.POOLER_TYPE)
            self.mask_head = build_mask_head(cfg, ShapeSpec(channels=c, width=m.POOLER_RESOLUTION,
                                                            height=m.POOLER_RESOLUTION))

    @torch.no_grad()
    def label_and_sample_proposals(self, proposals: List[Instances], targets: List[Instances]):
        from ..proposal_generator.proposal_utils import add_ground_truth_to_proposals
        from ..sampling import subsample_labels

        if self.proposal_append_gt:
            proposals = add_ground_truth_to_proposals([t.gt_boxes for t in targets], proposals)
        out = []
        for prop, tgt in zip(proposals, targets):
            has_gt = len(tgt) > 0
            idx, lab = self.proposal_matcher(pairwise_iou(tgt.gt_boxes, prop.proposal_boxes))
            if has_gt:
                cls = tgt.gt_classes[idx].clone()
                cls[lab == 0] = self.num_classes
                cls[lab == -1] = -1
            else:
                cls = torch.zeros_like(idx) + self.num_classes
            pos, neg = subsample_labels(cls, self.batch_size_per_image, self.positive_fraction, self.num_classes)
            keep = torch.cat([pos, neg], dim=0)
            prop = prop[keep]
            prop.gt_classes = cls[keep]
            if has_gt:
                chosen = idx[keep]
                for name, value in tgt.get_fields().items():
                    if name.startswith("gt_") and not prop.has(name):
                        prop.set(name, value[chosen])
                prop.set("matched_gt_idx", chosen)
            out.append(prop)
        return out

    def forward(self, images, features, proposals, targets=None):
        if self.training:
            assert targets, "'targets' argument is required during training"
            proposals = self.label_and_sample_proposals(proposals, targets)
            losses = self._forward_box(features, proposals)
            if self.mask_on:
                losses.update(self._forward_mask(features, proposals, targets))
            return proposals, losses
        pred = self._forward_box(features, proposals)
        return self.forward_with_given_boxes(features, pred), {}

    def _forward_box(self, features, proposals):
        feats = [features[f] for f in self.in_features]
        x = self.box_head(self.box_pooler(feats, [p.proposal_boxes for p in proposals]))
        predictions = self.box_predictor(x)
        if self.training:
            return self.box_predictor.losses(predictions, proposals)
        return self.box_predictor.inference(predictions, proposals)[0]

    @torch.no_grad()
    def forward_with_given_boxes(self, features, instances):
        if self.mask_on:
            from .mask_head import mask_rcnn_inference
            feats = [features[f] for f in self.in_features]
            x = self.mask_pooler(feats, [i.pred_boxes for i in instances])
            mask_rcnn_inference(self.mask_head.layers(x)[0], instances)
        return instances

    def _forward_mask(self, features, proposals, targets):
        """Foreground rois only; the target of a roi is its matched instance's bitmask cropped to the roi at
        2 x POOLER_RESOLUTION (crop_and_resize = ROIAlign(1.0, sampling 0, aligned) >= 0.5)."""
        from ...layers.roi_align import roi_align
        from .mask_head import mask_rcnn_loss

        fg, _ = select_foreground_proposals(proposals, self.num_classes)
        feats = [features[f] for f in self.in_features]
        logits, _ = self.mask_head.layers(self.mask_pooler(feats, [p.proposal_boxes for p in fg]))
        side = logits.shape[-1] if logits.shape[0] else 2 * self.mask_pooler.output_size[0]
        tgt, base = [], 0
        for p, t in zip(fg, targets):
            if len(p) == 0:
                continue
            if not isinstance(t.gt_masks, torch.Tensor) or t.gt_masks.dim() != 3:
                # (the reference's default INPUT.MASK_FORMAT is "polygon": PolygonMasks rasterised per roi by
                # pycocotools — outside the HIP scope; BitMasks users pass `.tensor`)
                raise TypeError("StandardROIHeads mask branch: gt_masks must be a (G, H, W) bitmask tensor "
                                "(INPUT.MASK_FORMAT = 'bitmask'), got %r" % type(t.gt_masks).__name__)
            masks = t.gt_masks.to(torch.float32)[:, None]                         # (G, 1, H, W) bitmasks
            rois = torch.cat([p.matched_gt_idx.to(torch.float32)[:, None], p.proposal_boxes.tensor], dim=1)
            tgt.append(roi_align(masks, rois, (side, side), 1.0, 0, True)[:, 0] >= 0.5)
        classes = torch.cat([p.gt_classes for p in fg]) if fg else logits.new_zeros(0, dtype=torch.long)
        if not tgt:
            return {"loss_mask": logits.sum() * 0}
        return {"loss_mask": mask_rcnn_loss(logits, classes.to(torch.int64), torch.cat(tgt))}
