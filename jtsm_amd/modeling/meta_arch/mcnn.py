"""GeneralizedMCNNWSL — projects/WSL/wsl/modeling/meta_arch/mcnn.py: the training forward :157-234
(+ preprocess_image :303-318): backbone -> roi_heads(images, features, proposals, gt, sem_seg,
superpixels) -> sem_seg_head(features, roi_heads.pgt_sem_seg); and inference :236-301 with its
post-processing (_postprocess :320-336, _postprocess_ps :338-365).

Input contract (mcnn.py:163-171,185-207): a list of dicts with `image` (3,H,W), `instances`
(gt_classes[, gt_boxes]), `sem_seg` (H,W) long, `proposals` Instances(proposal_boxes, objectness_logits,
oh_labels), `superpixels` (H,W) int.  Images are kept channels_last from the first pad onward.
"""
import os

import torch
from torch import nn

from ...structures import ImageList
from ..backbone import build_backbone
from ..roi_heads import build_roi_heads
from .build import META_ARCH_REGISTRY
from .semantic_seg import build_sem_seg_head
from ...layers.conv import planes_clear, set_segment
from ...layers.grad_fan import fan_out
from ...layers.postprocess import argmax_channels, preprocess_images_u8
from ..postprocessing import detector_postprocess, sem_seg_postprocess
from .panoptic_fpn import combine_semantic_and_instance_outputs

# The semantic head's forward enqueued inside the mask branch's synchronisation window (roi_heads_jtsm.py: sync_window):
# JTSM_SEM_IN_SYNC_WINDOW=0 restores the order of the reference's forward (heads one after the other).
SEM_IN_SYNC_WINDOW = os.environ.get("JTSM_SEM_IN_SYNC_WINDOW", "1") != "0"
# The semantic head on a side stream (JTSM_SEM_SIDE_STREAM=0 switches it off): its layers start as soon as the pyramid
# exists and run beside the box / mask branches; autograd runs a node's backward on the stream its forward ran on, so the
# head's backward runs beside theirs too.  Its GroupNorm / up-sampling / cross-entropy kernels are bound by HBM and vector
# issue, the branches' large contractions by the matrix pipes: same-process A/B 21.2-21.6 -> 20.7-20.9 ms per step
# (tools/sweeps/sem_side_ab.py).  The head's gradient for the pyramid reaches the shared maps through autograd's
# accumulation instead of layers/grad_fan.py, i.e. in another summation order: losses differ from the one-stream step in
# the last bits (6.7e-5 absolute on a loss of 10 after 30 optimizer steps), and are reproducible bit for bit among
# themselves (tools/sweeps/trajectory_soak.py: 10 trajectories of 30 steps; tools/sweeps/stream_soak.py: 52 steps).
# Round 4 first parked this switch: the step was not reproducible.  Cause (DESIGN §5 "the side-stream anomaly"): the
# bilinear up-sampling kernel, compiled with packed fp32 arithmetic, stored sums lacking one of their four terms in the
# last sixteen lanes of ~1 % of its wavefronts while a second hardware queue was busy — its operands, its registers and a
# second evaluation inside the kernel all correct.  Built without packed arithmetic (jtsm_amd/build.py: FILE_FLAGS) the
# effect is gone: 0 of 64 runs against 10-11 of 12.
SEM_SIDE_STREAM = os.environ.get("JTSM_SEM_SIDE_STREAM", "1") != "0"


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
        self.ps_on = cfg.WSL.PS_ON
        combine = cfg.MODEL.PANOPTIC_FPN.COMBINE
        self.combine_on = combine.ENABLED
        self.combine_overlap_threshold = combine.OVERLAP_THRESH
        self.combine_stuff_area_limit = combine.STUFF_AREA_LIMIT
        self.combine_instances_confidence_threshold = combine.INSTANCES_CONFIDENCE_THRESH

    @property
    def device(self):
        return self.pixel_mean.device

    def preprocess_image(self, batched_inputs):
        images = [x["image"].to(self.device, non_blocking=True) for x in batched_inputs]
        if images[0].is_cuda and images[0].dim() == 3 and images[0].shape[0] in (1, 3) and \
                all(im.dtype == images[0].dtype for im in images) and images[0].dtype in (torch.uint8, torch.float32):
            # the mappers' uint8 planes (or float32 images): normalise + pad + channels-last in one launch
            # (csrc/postprocess.hip), the very bits of `(x - mean) / std` + ImageList.from_tensors
            mean, std = self._pixel_stats()
            tensor, sizes = preprocess_images_u8(images, mean, std, self.backbone.size_divisibility)
            return ImageList(tensor, sizes)
        images = [(x - self.pixel_mean) / self.pixel_std for x in images]
        return ImageList.from_tensors(images, self.backbone.size_divisibility, channels_last=True)

    def _pixel_stats(self):
        """Host copies of the normalisation constants (read from the buffers once, then cached)."""
        cached = getattr(self, "_pixel_stats_cache", None)
        if cached is None:
            cached = self._pixel_stats_cache = (self.pixel_mean.flatten().tolist(), self.pixel_std.flatten().tolist())
        return cached

    def forward(self, batched_inputs):
        if not self.training:
            return self.inference(batched_inputs)
        planes_clear()   # bf16x3 operand planes are cached per step (layers/conv.py)
        images = self.preprocess_image(batched_inputs)
        gt_instances = [x["instances"].to(self.device) for x in batched_inputs]
        gt_sem_seg = ImageList.from_tensors([x["sem_seg"].to(self.device) for x in batched_inputs],
                                            self.backbone.size_divisibility, self.sem_seg_head.ignore_value).tensor
        set_segment("backbone")
        features = self.backbone(images.tensor)
        set_segment("heads")
        superpixels = ImageList.from_tensors([x["superpixels"].to(self.device) for x in batched_inputs],
                                             self.backbone.size_divisibility)
        proposals = [x["proposals"].to(self.device) for x in batched_inputs]
        # every FPN level has three readers (box pooler, mask pooler, semantic head): each gets its own view, through
        # which their backward kernels add into ONE gradient map per level instead of autograd adding three
        # (layers/grad_fan.py); without it, or for a head that does not take part, the views behave like `features`
        # (under the gradient exchange too: its collectives wait for every registered producer stream, engine/dp.py)
        use_side = SEM_SIDE_STREAM and images.tensor.is_cuda and hasattr(self.sem_seg_head, "layers")
        fans = {k: fan_out(v, 2 if use_side else 3) for k, v in features.items()}
        f_box, f_mask = ({k: v[i] for k, v in fans.items()} for i in range(2))
        f_sem = None if use_side else {k: v[2] for k, v in fans.items()}
        side = None
        if use_side:
            side = self._semantic_stream(images.tensor.device)
            main = torch.cuda.current_stream(images.tensor.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                # (the pyramid itself, not the fan views: the shared gradient maps of layers/grad_fan.py are added into in
                # place by the compute stream's nodes, which a node on another stream must not join — autograd's own
                # accumulation orders the two streams)
                sem_pred = self.sem_seg_head.layers(features)
        in_window = side is None and SEM_IN_SYNC_WINDOW and getattr(self.roi_heads, "mask_on", False) and \
            getattr(self.roi_heads, "takes_mask_features", False)
        if in_window:   # the semantic head's forward fills the wait for the mask branch's foreground count
            self.roi_heads.sync_window = lambda: self.sem_seg_head(f_sem, self.roi_heads.pgt_sem_seg)
        if getattr(self.roi_heads, "takes_mask_features", False):
            _, detector_losses = self.roi_heads(images, f_box, proposals, gt_instances, gt_sem_seg, superpixels,
                                                mask_features=f_mask)
        else:
            _, detector_losses = self.roi_heads(images, f_box, proposals, gt_instances, gt_sem_seg, superpixels)
        if side is not None:
            main = torch.cuda.current_stream(images.tensor.device)
            side.wait_stream(main)                    # the pseudo semantic target is painted on the compute stream
            with torch.cuda.stream(side):
                sem_seg_losses = self.sem_seg_head.losses(sem_pred, self.roi_heads.pgt_sem_seg)
            main.wait_stream(side)                    # the caller sums the losses on the compute stream
        elif in_window:
            self.roi_heads.sync_window = None
            _, sem_seg_losses = self.roi_heads.sync_window_result
            self.roi_heads.sync_window_result = None
        else:
            _, sem_seg_losses = self.sem_seg_head(f_sem, self.roi_heads.pgt_sem_seg)
        losses = {}
        losses.update(sem_seg_losses)
        losses.update(detector_losses)
        return losses

    def _semantic_stream(self, device):
        side = getattr(self, "_sem_stream", None)
        if side is None:
            side = self._sem_stream = torch.cuda.Stream(device=device)
            from ...layers.conv import register_producer_stream
            register_producer_stream(side)      # (engine/dp.py: the exchange's collectives wait for it)
        return side

    @torch.no_grad()
    def inference(self, batched_inputs, detected_instances=None, do_postprocess=True, only_sem_seg=False):
        """mcnn.py:236-301.  Returns, per image, {"instances", "sem_seg"[, "panoptic_seg"]} when PS_ON, else
        {"instances"}; with do_postprocess=False the raw (results, all_scores, all_boxes)."""
        assert not self.training
        planes_clear()
        images = self.preprocess_image(batched_inputs)
        features = self.backbone(images.tensor)
        assert "superpixels" in batched_inputs[0] and "proposals" in batched_inputs[0]
        superpixels = ImageList.from_tensors([x["superpixels"].to(self.device) for x in batched_inputs],
                                             self.backbone.size_divisibility)
        proposals = [x["proposals"].to(self.device) for x in batched_inputs]
        if only_sem_seg:
            sem_seg_results, _ = self.sem_seg_head(features, None)
            return sem_seg_results, None, None
        # the semantic head beside the box / mask branches, as in training (it depends on the pyramid only)
        side = self._semantic_stream(images.tensor.device) if SEM_SIDE_STREAM and images.tensor.is_cuda else None
        if side is not None:
            main = torch.cuda.current_stream(images.tensor.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                sem_seg_results, _ = self.sem_seg_head(features, None)
        if detected_instances is None:
            results, _, all_scores, all_boxes = self.roi_heads(images, features, proposals, None, None, superpixels)
        else:
            detected_instances = [x.to(self.device) for x in detected_instances]
            self.roi_heads.proposals, self.roi_heads.superpixels, self.roi_heads.images = proposals, superpixels, images
            results, all_scores, all_boxes = self.roi_heads.forward_with_given_boxes(features, detected_instances)
        if side is not None:
            main.wait_stream(side)
            for t in (sem_seg_results if isinstance(sem_seg_results, (list, tuple)) else [sem_seg_results]):
                if isinstance(t, torch.Tensor):
                    t.record_stream(main)       # (allocated under the side stream, read and freed on the compute stream)
        else:
            sem_seg_results, _ = self.sem_seg_head(features, None)
        if do_postprocess and self.ps_on:
            return self._postprocess_ps(sem_seg_results, results, batched_inputs, images.image_sizes)
        if do_postprocess:
            return GeneralizedMCNNWSL._postprocess(results, batched_inputs, images.image_sizes)
        return results, all_scores, all_boxes

    @staticmethod
    def _postprocess(instances, batched_inputs, image_sizes):
        processed_results = []
        for results_per_image, input_per_image, image_size in zip(instances, batched_inputs, image_sizes):
            height = input_per_image.get("height", image_size[0])
            width = input_per_image.get("width", image_size[1])
            processed_results.append({"instances": detector_postprocess(results_per_image, height, width)})
        return processed_results

    def _postprocess_ps(self, sem_seg_results, detector_results, batched_inputs, image_sizes):
        processed_results = []
        for sem_seg_result, detector_result, input_per_image, image_size in zip(sem_seg_results, detector_results,
                                                                              batched_inputs, image_sizes):
            height = input_per_image.get("height", image_size[0])
            width = input_per_image.get("width", image_size[1])
            sem_seg_r = sem_seg_postprocess(sem_seg_result, image_size, height, width)
            detector_r = detector_postprocess(detector_result, height, width)
            processed_results.append({"sem_seg": sem_seg_r, "instances": detector_r})
            if self.combine_on:
                processed_results[-1]["panoptic_seg"] = combine_semantic_and_instance_outputs(
                    detector_r, argmax_channels(sem_seg_r), self.combine_overlap_threshold,
                    self.combine_stuff_area_limit, self.combine_instances_confidence_threshold,
                    num_sem_classes=sem_seg_r.shape[0])
        return processed_results
