"""GeneralizedRCNN and PanopticFPN — call surface of detectron2/modeling/meta_arch/rcnn.py:23-246 and
panoptic_fpn.py:20-130: backbone -> proposal generator (RPN, or the dataset's proposals) -> ROI heads (-> semantic head
and the panoptic merge).  BASELINE configs[0] (configs/COCO-Detection/faster_rcnn_R_50_FPN_1x.yaml) builds and trains
through this; north_star asks that the `GeneralizedRCNN / build_model()` registry keeps working.  Every convolution /
linear layer, ROIAlign, NMS and the post-processing run on this repo's HIP kernels; the glue is device-side torch."""
import torch
from torch import nn

from ...layers.conv import planes_clear
from ...layers.postprocess import argmax_channels
from ...structures import ImageList
from ..backbone import build_backbone
from ..postprocessing import detector_postprocess, sem_seg_postprocess
from ..proposal_generator import build_proposal_generator
from ..roi_heads import build_roi_heads
from .build import META_ARCH_REGISTRY
from .panoptic_fpn import combine_semantic_and_instance_outputs
from .semantic_seg import build_sem_seg_head


class _RCNNBase(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.backbone = build_backbone(cfg)
        self.proposal_generator = build_proposal_generator(cfg, self.backbone.output_shape())
        self.roi_heads = build_roi_heads(cfg, self.backbone.output_shape())
        assert len(cfg.MODEL.PIXEL_MEAN) == len(cfg.MODEL.PIXEL_STD)
        self.register_buffer("pixel_mean", torch.Tensor(cfg.MODEL.PIXEL_MEAN).view(-1, 1, 1))
        self.register_buffer("pixel_std", torch.Tensor(cfg.MODEL.PIXEL_STD).view(-1, 1, 1))
        self.input_format = cfg.INPUT.FORMAT

    @property
    def device(self):
        return self.pixel_mean.device

    def preprocess_image(self, batched_inputs):
        images = [(x["image"].to(self.device) - self.pixel_mean) / self.pixel_std for x in batched_inputs]
        return ImageList.from_tensors(images, self.backbone.size_divisibility, channels_last=True)

    def _proposals(self, images, features, batched_inputs, gt_instances):
        if self.proposal_generator is not None:
            return self.proposal_generator(images, features, gt_instances)
        assert "proposals" in batched_inputs[0], "no proposal generator and no precomputed proposals"
        return [x["proposals"].to(self.device) for x in batched_inputs], {}


@META_ARCH_REGISTRY.register()
class GeneralizedRCNN(_RCNNBase):
    def forward(self, batched_inputs):
        if not self.training:
            return self.inference(batched_inputs)
        planes_clear()
        images = self.preprocess_image(batched_inputs)
        gt_instances = [x["instances"].to(self.device) for x in batched_inputs] if "instances" in batched_inputs[0] else None
        features = self.backbone(images.tensor)
        proposals, proposal_losses = self._proposals(images, features, batched_inputs, gt_instances)
        _, detector_losses = self.roi_heads(images, features, proposals, gt_instances)
        losses = {}
        losses.update(detector_losses)
        losses.update(proposal_losses)
        return losses

    @torch.no_grad()
    def inference(self, batched_inputs, detected_instances=None, do_postprocess=True):
        assert not self.training
        planes_clear()
        images = self.preprocess_image(batched_inputs)
        features = self.backbone(images.tensor)
        if detected_instances is None:
            proposals, _ = self._proposals(images, features, batched_inputs, None)
            results, _ = self.roi_heads(images, features, proposals, None)
        else:
            results = self.roi_heads.forward_with_given_boxes(features, [x.to(self.device) for x in detected_instances])
        if not do_postprocess:
            return results
        out = []
        for r, inp, size in zip(results, batched_inputs, images.image_sizes):
            out.append({"instances": detector_postprocess(r, inp.get("height", size[0]), inp.get("width", size[1]))})
        return out


@META_ARCH_REGISTRY.register()
class PanopticFPN(_RCNNBase):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.sem_seg_head = build_sem_seg_head(cfg, self.backbone.output_shape())
        p = cfg.MODEL.PANOPTIC_FPN
        self.instance_loss_weight = p.INSTANCE_LOSS_WEIGHT
        self.combine_on = p.COMBINE.ENABLED
        self.combine_overlap_threshold = p.COMBINE.OVERLAP_THRESH
        self.combine_stuff_area_limit = p.COMBINE.STUFF_AREA_LIMIT
        self.combine_instances_confidence_threshold = p.COMBINE.INSTANCES_CONFIDENCE_THRESH

    def forward(self, batched_inputs):
        planes_clear()
        images = self.preprocess_image(batched_inputs)
        features = self.backbone(images.tensor)
        gt_sem_seg = None
        if "sem_seg" in batched_inputs[0]:
            gt_sem_seg = ImageList.from_tensors([x["sem_seg"].to(self.device) for x in batched_inputs],
                                                self.backbone.size_divisibility, self.sem_seg_head.ignore_value).tensor
        sem_seg_results, sem_seg_losses = self.sem_seg_head(features, gt_sem_seg)
        gt_instances = [x["instances"].to(self.device) for x in batched_inputs] if "instances" in batched_inputs[0] else None
        proposals, proposal_losses = self._proposals(images, features, batched_inputs, gt_instances)
        detector_results, detector_losses = self.roi_heads(images, features, proposals, gt_instances)
        if self.training:
            losses = dict(sem_seg_losses)
            losses.update({k: v * self.instance_loss_weight for k, v in detector_losses.items()})
            losses.update(proposal_losses)
            return losses
        out = []
        for sem, det, inp, size in zip(sem_seg_results, detector_results, batched_inputs, images.image_sizes):
            h, w = inp.get("height", size[0]), inp.get("width", size[1])
            sem_r, det_r = sem_seg_postprocess(sem, size, h, w), detector_postprocess(det, h, w)
            out.append({"sem_seg": sem_r, "instances": det_r})
            if self.combine_on:
                out[-1]["panoptic_seg"] = combine_semantic_and_instance_outputs(
                    det_r, argmax_channels(sem_r), self.combine_overlap_threshold, self.combine_stuff_area_limit,
                    self.combine_instances_confidence_threshold, num_sem_classes=sem_r.shape[0])
        return out
