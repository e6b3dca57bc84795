"""GeneralizedMCNNWSL — the training forward of projects/WSL/wsl/modeling/meta_arch/mcnn.py:157-234
(+ preprocess_image :303-318): backbone -> roi_heads(images, features, proposals, gt, sem_seg,
superpixels) -> sem_seg_head(features, roi_heads.pgt_sem_seg).

Input contract (mcnn.py:163-171,185-207): a list of dicts with `image` (3,H,W), `instances`
(gt_classes[, gt_boxes]), `sem_seg` (H,W) long, `proposals` Instances(proposal_boxes, objectness_logits,
oh_labels), `superpixels` (H,W) int.  Images are kept channels_last from the first pad onward.
"""
import torch
from torch import nn

from ...structures import ImageList
from ..backbone import build_backbone
from ..roi_heads import build_roi_heads
from .build import META_ARCH_REGISTRY
from .semantic_seg import build_sem_seg_head
from ...layers.conv import planes_clear


@META_ARCH_REGISTRY.register()
class GeneralizedMCNNWSL(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.backbone = build_backbone(cfg)
        if cfg.MODEL.PROPOSAL_GENERATOR.NAME != "PrecomputedProposals":
            raise NotImplementedError("JTSM trains on precomputed proposals (proposal_generator/build.py:20-22)")
        self.proposal_generator = None
        self.roi_heads = build_roi_heads(cfg, self.backbone.output_shape())
        self.sem_seg_head = build_sem_seg_head(cfg, self.backbone.output_shape())
        assert len(cfg.MODEL.PIXEL_MEAN) == len(cfg.MODEL.PIXEL_STD)
        self.register_buffer("pixel_mean", torch.Tensor(cfg.MODEL.PIXEL_MEAN).view(-1, 1, 1))
        self.register_buffer("pixel_std", torch.Tensor(cfg.MODEL.PIXEL_STD).view(-1, 1, 1))
        self.input_format = cfg.INPUT.FORMAT

    @property
    def device(self):
        return self.pixel_mean.device

    def preprocess_image(self, batched_inputs):
        images = [x["image"].to(self.device, non_blocking=True) for x in batched_inputs]
        images = [(x - self.pixel_mean) / self.pixel_std for x in images]
        return ImageList.from_tensors(images, self.backbone.size_divisibility, channels_last=True)

    def forward(self, batched_inputs):
        if not self.training:
            raise NotImplementedError("jtsm_amd implements the training step; inference is a 'next' row (§8f)")
        planes_clear()   # bf16x3 operand planes are cached per step (layers/conv.py)
        images = self.preprocess_image(batched_inputs)
        gt_instances = [x["instances"].to(self.device) for x in batched_inputs]
        gt_sem_seg = ImageList.from_tensors([x["sem_seg"].to(self.device) for x in batched_inputs],
                                            self.backbone.size_divisibility, self.sem_seg_head.ignore_value).tensor
        features = self.backbone(images.tensor)
        superpixels = ImageList.from_tensors([x["superpixels"].to(self.device) for x in batched_inputs],
                                             self.backbone.size_divisibility)
        proposals = [x["proposals"].to(self.device) for x in batched_inputs]
        _, detector_losses = self.roi_heads(images, features, proposals, gt_instances, gt_sem_seg, superpixels)
        _, sem_seg_losses = self.sem_seg_head(features, self.roi_heads.pgt_sem_seg)
        losses = {}
        losses.update(sem_seg_losses)
        losses.update(detector_losses)
        return losses
