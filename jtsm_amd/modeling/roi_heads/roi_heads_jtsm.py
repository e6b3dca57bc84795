"""JTSMROIHeads — projects/WSL/wsl/modeling/roi_heads/roi_heads_jtsm.py: the training path (forward :502-552,
_forward_box :590-737, _forward_mask :754-948, get_pgt_top_k :1167-1338 (top_k = 1, thres = 0), get_pgt_sem_seg
:2025-2070, get_image_level_gt_stuff :165-194) and the inference path (forward :553-561, _forward_box :738-752,
forward_with_given_boxes :563-588, _forward_mask :949-961).

MI355X mapping of the box branch (the reference issues ~40 small launches + K+1 GEMM pairs here):
  MOIPool on all FPN levels, no host sync           -> 4 launches (+2 tiny bit-set builders each)
  mask_scale * (objectness + 1)                     -> one per-roi factor, one multiply
  DAN fc1 / fc2 (+bias +ReLU)                       -> 2 MFMA GEMMs
  cls, det and the K refinement heads (2K+2 Linears) -> ONE MFMA GEMM over the concatenated weights
  MIL scores + image probabilities + BCE (+backward) -> 3 (+1) launches
  each refinement's weighted CE + weighted L1        -> 2 (+1) launches
Pseudo-label mining (top-1 per present class, IoU matching) stays in torch ops on the device, as in
the reference; it is label generation, not differentiated.

Mask pseudo labels, all on the device and without host round trips:
  * the "10 nearest" targets (:840-905): per pseudo box the top-10 foreground proposals by IoU, then every
    foreground proposal takes the mask of its best-matching near target (csrc/mining.hip: near_targets);
  * that mask is the reference's own grabCut-free construction, SUPERPIXEL EVIDENCE (object_evidence :1928-1994):
    the union of the superpixels the target's oh_labels row marks, cropped to the proposal at 28x28
    (csrc/roi_align.hip: sp_mask_targets — the full-image mask is never rasterised);
  * the mask refinery trains on get_pgt_mask's paste -> crop of the first head's prediction (:1997-2022,
    csrc/roi_align.hip: paste_crop_targets).
Declared substitutions (identical in oracle/model.py): grabCut itself (the reference's live branch of
object_evidence, an OpenCV CPU routine) is replaced by the superpixel evidence above; masks stay bitmasks where the
reference encodes them as polygons and rasterises those again (cv2.findContours / pycocotools).  The pseudo SEMANTIC
target is painted from the same superpixel-evidence masks, as the reference paints it from its targets' pgt_masks
(:2038-2069).  mask_targets / sem_targets = "rect" restore round 1's rectangles shrunk by 2 px.
"""
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from ...layers.conv import linear_fused, linear_fused_split
from ...layers.grad_fan import fan_out
from ...layers.mining import (MAX_IMAGES, fg_compact, image_labels, match_label, mine_top1, near_targets, pad_class_lists,
                              paint_sem_seg, paint_sem_seg_evidence, paste_crop_targets, rect_mask_targets, row_lse,
                              sp_mask_targets)
from ...layers.mining import roi_scale as fused_roi_scale
from ...layers.roi_align import roi_align
from ...layers.shape_spec import ShapeSpec
from ...structures import Boxes, ImageList, Instances
from ..poolers import ROIPooler, moi_label_inputs
from .box_head import build_box_head
from .fast_rcnn_oicr import OICROutputLayers
from .fast_rcnn_tsm import TSMOutputLayers
from .mask_head import build_mask_head, mask_rcnn_inference, mask_rcnn_loss
from .roi_heads import ROI_HEADS_REGISTRY, ROIHeads, get_image_level_gt, select_foreground_proposals


@torch.no_grad()
def get_image_level_gt_stuff(gt_sem_seg, num_classes_stuff, offset):
    """Stuff classes present per image: unique(sem_seg) minus {0 (things), 255 (ignore)}, shifted to
    [offset, offset + num_classes_stuff - 1); also their one-hot (B, num_classes_stuff - 1)."""
    if gt_sem_seg is None:
        return None, None, None
    present = []
    for t in gt_sem_seg:
        u = torch.unique(t, sorted=True)
        present.append((u[(u != 255) & (u != 0)] - 1).to(torch.int64))
    oh = torch.zeros((len(present), num_classes_stuff - 1), dtype=torch.float, device=gt_sem_seg.device)
    for i, u in enumerate(present):
        oh[i, u] = 1
    shifted = [u + offset for u in present]
    return shifted, shifted, oh


def eroded_rect_masks(boxes, height, width, erode=2.0):
    """(G,H,W) float 0/1 masks: pixel centres inside the box shrunk by `erode` px on every side."""
    ys = torch.arange(height, device=boxes.device, dtype=boxes.dtype).view(1, height, 1) + 0.5
    xs = torch.arange(width, device=boxes.device, dtype=boxes.dtype).view(1, 1, width) + 0.5
    b = boxes.view(-1, 4, 1, 1)
    return ((xs >= b[:, 0] + erode) & (xs <= b[:, 2] - erode) & (ys >= b[:, 1] + erode) &
            (ys <= b[:, 3] - erode)).to(torch.float32)


@torch.no_grad()
def present_things(targets, num_classes):
    """(B, num_classes) 0/1: which thing classes each image's instances contain (no host synchronisation)."""
    oh = torch.zeros((len(targets), num_classes), dtype=torch.float32, device=targets[0].gt_classes.device)
    for i, t in enumerate(targets):
        oh[i, t.gt_classes.to(torch.int64)] = 1
    return oh


@torch.no_grad()
def present_stuff(gt_sem_seg, num_classes_stuff):
    """(B, num_classes_stuff - 1) 0/1: which stuff labels 1 .. num_classes_stuff-1 occur in each (B,H,W) map
    (0 = things and 255 = ignore do not count) — one scatter instead of a sort-based unique per image."""
    b = gt_sem_seg.shape[0]
    flags = torch.zeros((b, 256), dtype=torch.float32, device=gt_sem_seg.device)
    flags.scatter_(1, gt_sem_seg.reshape(b, -1).clamp(0, 255), 1.0)
    return flags[:, 1:num_classes_stuff].contiguous()


@torch.no_grad()
def class_lists(oh, offset=0):
    """Padded per-image class lists of a (B,C) presence matrix: classes (B,C) int32 — the present ones first, in
    ascending order, each shifted by `offset` — and counts (B,) int32.  Lengths stay on the device."""
    c = oh.shape[1]
    ar = torch.arange(c, device=oh.device)
    key = torch.where(oh > 0, ar, ar + c)
    cls = (torch.sort(key, dim=1).values % c + offset).to(torch.int32)
    return cls.contiguous(), oh.sum(dim=1).to(torch.int32)


@ROI_HEADS_REGISTRY.register()
class JTSMROIHeads(ROIHeads):
    def __init__(self, cfg, input_shape: Dict[str, ShapeSpec]):
        super().__init__(**ROIHeads.from_config(cfg))
        self.num_classes_stuff = cfg.MODEL.SEM_SEG_HEAD.NUM_CLASSES
        self.mask_on = cfg.MODEL.MASK_ON
        self.refine_K = cfg.WSL.REFINE_NUM
        self.refine_reg = cfg.WSL.REFINE_REG
        self.cls_agnostic_bbox_reg = cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG
        assert not self.cls_agnostic_bbox_reg and not cfg.WSL.REFINE_MIST

        # ---- box branch (roi_heads_jtsm.py:347-404)
        in_features = cfg.MODEL.ROI_HEADS.IN_FEATURES
        self.box_in_features = self.mask_in_features = in_features
        scales = tuple(1.0 / input_shape[k].stride for k in in_features)
        in_channels = [input_shape[f].channels for f in in_features]
        assert len(set(in_channels)) == 1, in_channels
        in_channels = in_channels[0]
        res = cfg.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION
        self.box_pooler = ROIPooler(output_size=res, scales=scales,
                                    sampling_ratio=cfg.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO,
                                    pooler_type=cfg.MODEL.ROI_BOX_HEAD.POOLER_TYPE)
        self.box_head = build_box_head(cfg, ShapeSpec(channels=in_channels, height=res, width=res))
        feat = self.box_head.output_shape.channels
        self.box_predictor = TSMOutputLayers.from_config(cfg, feat)
        self.box_refinery = []
        for k in range(self.refine_K):
            layer = OICROutputLayers.from_config(cfg, feat, k)
            self.add_module("box_refinery_{}".format(k), layer)
            self.box_refinery.append(layer)

        # ---- mask branch (roi_heads_jtsm.py:406-466): two class-specific heads
        if self.mask_on:
            mres = cfg.MODEL.ROI_MASK_HEAD.POOLER_RESOLUTION
            self.mask_pooler = ROIPooler(output_size=mres, scales=scales,
                                         sampling_ratio=cfg.MODEL.ROI_MASK_HEAD.POOLER_SAMPLING_RATIO,
                                         pooler_type=cfg.MODEL.ROI_MASK_HEAD.POOLER_TYPE)
            shape = ShapeSpec(channels=in_channels, width=mres, height=mres)
            self.mask_head = build_mask_head(cfg, shape)
            self.mask_refinery = []
            head = build_mask_head(cfg, shape)
            self.add_module("mask_refinery_0", head)
            self.mask_refinery.append(head)
            for h in [self.mask_head] + self.mask_refinery:
                h.return_features = False    # only the logits of `layers` are used here: no fp32 upsampled features
        self.pgt_sem_seg = None
        self.aux = {}
        self.mask_mined_top_k = 10            # roi_heads_jtsm.py:420 (self.mask_mined_top_k = 10)
        self.mask_targets = "evidence"        # "evidence" (reference semantics) or "rect" (round-1 rectangles)
        self.sem_targets = "evidence"         # pseudo semantic target: from the evidence masks (:2038-2069) or "rect"
        self._evidence = None                 # (oh_labels of all proposals (R,L) int32, superpixels (B,H,W) int32)

    # ------------------------------------------------------------------ label mining (no grad)
    @torch.no_grad()
    def get_pgt_top_k(self, prev_pred_boxes, prev_pred_scores, proposals, num_classes, gt_classes_img_int):
        """One pseudo ground-truth box per present class: the highest-scoring proposal of that class
        column; its weight is the image-level score of the class.  Returns list[Instances] with
        gt_boxes, gt_classes, gt_scores, gt_weights, oh_labels, and the winning row (pgt_idx)."""
        targets = []
        for i, (boxes_i, scores_i, prop_i, cls_i) in enumerate(zip(prev_pred_boxes, prev_pred_scores, proposals,
                                                                  gt_classes_img_int)):
            if isinstance(boxes_i, Boxes):
                boxes_i = boxes_i.tensor.unsqueeze(1).expand(len(boxes_i), num_classes, 4)
            boxes_i = boxes_i.reshape(-1, num_classes, 4)
            sc = torch.index_select(scores_i, 1, cls_i)                       # (R_i, G)
            top, idx = torch.topk(sc, min(sc.size(0), 1), dim=0)            # (1, G)
            bx = torch.index_select(boxes_i, 1, cls_i)                        # (R_i, G, 4)
            picked = torch.gather(bx, 0, idx.unsqueeze(2).expand(-1, -1, 4)).reshape(-1, 4)
            weights = torch.index_select(self.pred_class_img_logits[i:i + 1], 1, cls_i).reshape(-1)
            targets.append(Instances(prop_i.image_size, gt_boxes=Boxes(picked), gt_classes=cls_i.clone(),
                                     gt_scores=top.reshape(-1), gt_weights=weights,
                                     oh_labels=prop_i.oh_labels[idx.reshape(-1)], pgt_idx=idx.reshape(-1)))
        return targets

    @torch.no_grad()
    def get_pgt_sem_seg(self, prev_pred_boxes, prev_pred_scores, proposals, height, width):
        """Reference-shaped entry (list-of-Instances mining, roi_heads_jtsm.py:2025-2070): the targets' masks are
        their superpixel evidence (object_evidence :1928-1994) — or, with sem_targets == "rect", their rectangles
        shrunk by 2 px."""
        stuff_lists = [c[:int(n)].to(torch.int64) for c, n in zip(self.stuff_cls, self.stuff_cnt.tolist())]   # (syncs)
        targets = self.get_pgt_top_k(prev_pred_boxes, prev_pred_scores, proposals,
                                     self.num_classes + self.num_classes_stuff - 1, stuff_lists)
        if self.sem_targets == "evidence":
            sp = self.superpixels.tensor if hasattr(self.superpixels, "tensor") else self.superpixels
            masks = []
            for t, sp_i in zip(targets, sp):
                ids = sp_i.reshape(1, -1).to(torch.int64)
                width_l = t.oh_labels.shape[1]
                inside = (ids >= 0) & (ids < width_l)
                hit = torch.gather(t.oh_labels.to(torch.int64), 1, ids.clamp(0, width_l - 1).expand(len(t), -1)) != 0
                masks.append((hit & inside).reshape(len(t), *sp_i.shape))
        else:
            masks = [eroded_rect_masks(t.gt_boxes.tensor, height, width) > 0.5 for t in targets]
        return self._paint_sem_seg(masks, [t.gt_classes for t in targets], [t.gt_scores for t in targets], height, width)

    @torch.no_grad()
    def _paint_sem_seg(self, masks, classes, scores, height, width):
        """Pseudo semantic target: every pseudo stuff target paints its (n,H,W) bool mask with its class id, in
        ascending score order so the best target ends on top; a class that got painted over completely is painted
        once more (in class order).  Written without host synchronisation: "ascending order, last write wins" is the
        per-pixel maximum of the targets' score RANKS."""
        out = torch.zeros(len(masks), height, width, device=masks[0].device, dtype=torch.int64)
        for i, (mk, cl, sc) in enumerate(zip(masks, classes, scores)):
            n = mk.shape[0]
            if n == 0:
                continue
            vals = cl.to(torch.int64) - self.num_classes + 1
            order = torch.argsort(sc, descending=False)
            rank = torch.empty_like(order)
            rank[order] = torch.arange(n, device=order.device)
            top = torch.where(mk, rank.view(n, 1, 1), rank.new_full((), -1)).max(dim=0).values
            painted = vals[order[top.clamp(min=0)]]
            img = torch.where(top >= 0, painted, painted.new_zeros(()))
            for j in range(n):                                                  # sequential, but sync-free
                missing = ~(img == vals[j]).any()
                img = torch.where(missing & mk[j], vals[j], img)
            out[i] = img
        return out

    # ------------------------------------------------------------------ forward
    takes_mask_features = True     # (forward's `mask_features`: the mask pooler's own views of the feature levels)

    def forward(self, images: ImageList, features: Dict[str, torch.Tensor], proposals: List[Instances],
                targets: Optional[List[Instances]] = None, gt_sem_seg: Optional[torch.Tensor] = None,
                superpixels: ImageList = None, mask_features: Optional[Dict[str, torch.Tensor]] = None):
        if not self.training:
            # roi_heads_jtsm.py:553-561: K-head averaged detections, then the mask heads on the detected boxes
            self.proposals, self.superpixels, self.images = proposals, superpixels, images
            pred_instances, all_scores, all_boxes = self._forward_box_inference(features, proposals)
            pred_instances, _, _ = self.forward_with_given_boxes(features, pred_instances)
            return pred_instances, {}, all_scores, all_boxes
        assert targets, "'targets' argument is required during training"
        self.proposals, self.superpixels, self.images = proposals, superpixels, images
        # image-level labels, entirely on the device (presence matrices + padded class lists with counts)
        self.has_stuff = gt_sem_seg is not None
        if (len(targets) <= MAX_IMAGES and targets[0].gt_classes.is_cuda and self.num_classes <= 256 and
                (not self.has_stuff or (gt_sem_seg.dim() == 3 and self.num_classes_stuff <= 256))):
            # three launches (layers/mining.py: image_labels); the tensor-op helpers below define what it returns
            (self.gt_classes_img_oh, self.things_cls, self.things_cnt, oh_s, cls_s, cnt_s) = image_labels(
                [t.gt_classes for t in targets], self.num_classes, gt_sem_seg, self.num_classes_stuff, self.num_classes)
            if self.has_stuff:
                self.gt_classes_img_oh_stuff, self.stuff_cls, self.stuff_cnt = oh_s, cls_s, cnt_s
        else:
            self.gt_classes_img_oh = present_things(targets, self.num_classes)
            self.things_cls, self.things_cnt = class_lists(self.gt_classes_img_oh)
            if self.has_stuff:
                self.gt_classes_img_oh_stuff = present_stuff(gt_sem_seg, self.num_classes_stuff)
                self.stuff_cls, self.stuff_cnt = class_lists(self.gt_classes_img_oh_stuff, offset=self.num_classes)
        losses = self._forward_box(features, proposals)
        if self.mask_on:
            self._mask_prepare()
            # work that does not depend on the foreground count, enqueued while the count is on its way to the host (the
            # meta-architecture hands the semantic head's forward here): the device keeps busy behind the step's one
            # synchronisation instead of draining while the host waits
            window = getattr(self, "sync_window", None)
            self.sync_window_result = window() if window is not None else None
            losses.update(self._forward_mask(mask_features if mask_features is not None else features, proposals))
        return proposals, losses

    def _predictor_layers(self):
        mods = [self.box_predictor.cls, self.box_predictor.det]
        for r in self.box_refinery:
            mods += [r.cls_score] + ([r.bbox_pred] if r.has_reg else [])
        return mods

    def _predictor_gemm(self, x):
        """cls, det and every refinement head in one GEMM; returns the column slices."""
        mods = self._predictor_layers()
        return linear_fused_split(x, [m.weight for m in mods], [m.bias for m in mods])

    def _box_features(self, features, proposals):
        """MOIPool -> per-roi rescale -> DAN -> every predictor in one GEMM (roi_heads_jtsm.py:607-633)."""
        feats = [features[f] for f in self.box_in_features]
        # the label operands in kernel form, once: MOIPool reads them here, the evidence masks of the semantic target
        # and of the mask branch read the same tensors later (handed on explicitly, not through the pooler's state)
        self._evidence = moi_label_inputs([x.oh_labels for x in proposals], self.superpixels)
        box_features, argmax = self.box_pooler(feats, [x.proposal_boxes for x in proposals],
                                               oh_labels_list=self._evidence[0], superpixels=self._evidence[1])
        with torch.no_grad():
            if (argmax.is_cuda and argmax.dtype == torch.int32 and len(proposals) <= MAX_IMAGES and
                    argmax.is_contiguous(memory_format=torch.channels_last)):
                roi_scale = fused_roi_scale(argmax, [x.objectness_logits for x in proposals])   # one launch
            else:
                bins = argmax.size(2) * argmax.size(3)
                nvalid = (argmax[:, 0, :, :] != -1).reshape(argmax.size(0), -1).sum(dim=1).to(dtype=torch.float32)
                roi_scale = bins * (nvalid + 1).reciprocal()
                roi_scale = roi_scale * torch.cat([x.objectness_logits + 1 for x in proposals], dim=0)
        # reference: (features * mask_scale) * (objectness + 1), two passes; here one combined factor, which the box
        # head folds into the plane split in front of fc1 and into fc1's data-gradient epilogue (no multiply pass)
        if getattr(self.box_head, "takes_roi_scale", False):
            # (the predictors' GEMM rides in the box head's node when that is the fused stack: `tail`)
            mods = self._predictor_layers()
            res = self.box_head(box_features, roi_scale=roi_scale, tail=([m.weight for m in mods], [m.bias for m in mods]))
            if isinstance(res, tuple):
                return res[1], argmax
            box_features = res
        else:
            box_features = self.box_head(box_features * roi_scale.view(-1, 1, 1, 1))
        return self._predictor_gemm(box_features), argmax

    @torch.no_grad()
    def _forward_box_inference(self, features, proposals):
        """roi_heads_jtsm.py:736-752: every refinement head's prediction, averaged inside
        OICROutputLayers.inference."""
        outs, _ = self._box_features(features, proposals)
        if not all(r.has_reg for r in self.box_refinery):
            raise NotImplementedError("inference averages the box deltas of every refinement head "
                                      "(fast_rcnn_oicr.py:729-733): WSL.REFINE_REG must be all True")
        predictions_K, col = [], 2
        for _ in self.box_refinery:
            predictions_K.append((outs[col], outs[col + 1]))
            col += 2
        pred_instances, _, all_scores, all_boxes = self.box_refinery[-1].inference(predictions_K, proposals)
        return pred_instances, all_scores, all_boxes

    @torch.no_grad()
    def forward_with_given_boxes(self, features, instances):
        """Mask prediction for given `pred_boxes` / `pred_classes` (roi_heads_jtsm.py:563-588, 949-961): the logits
        of the refinement mask heads are averaged, then the class channel's sigmoid."""
        assert not self.training
        assert instances[0].has("pred_boxes") and instances[0].has("pred_classes")
        if self.mask_on:
            feats = [features[f] for f in self.mask_in_features]
            mask_features = self.mask_pooler(feats, [x.pred_boxes for x in instances])
            mask_rcnn_inference([head.layers(mask_features)[0] for head in self.mask_refinery], instances)
        return instances, [], []

    def _forward_box(self, features, proposals):
        counts = [len(p) for p in proposals]
        outs, argmax = self._box_features(features, proposals)
        dev = outs[0].device
        cls_logits, det_logits = outs[0], outs[1]
        offsets = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32).to(dev, non_blocking=True)
        labels_oh = (torch.cat([self.gt_classes_img_oh, self.gt_classes_img_oh_stuff], dim=1)
                     if self.has_stuff else self.gt_classes_img_oh)
        losses, scores, img_probs = self.box_predictor.score_and_loss(cls_logits, det_logits, offsets, labels_oh,
                                                                      max(counts))
        self.pred_class_img_logits = img_probs
        self.aux = {"mil_scores": scores, "img_probs": img_probs, "pooled_argmax": argmax}
        all_boxes = torch.cat([p.proposal_boxes.tensor for p in proposals]).contiguous()
        things_cls, things_cnt = self.things_cls, self.things_cnt
        self._mining = (all_boxes, offsets, things_cls, things_cnt, counts)

        # pseudo semantic target from the top-1 box of every present stuff class (MIL scores), painted by
        # libjtsm_hip.so (csrc/mining.hip: paint_*), list lengths never leaving the device
        if self.has_stuff:
            pg = mine_top1(scores, all_boxes, offsets, self.stuff_cls, self.stuff_cnt, img_probs)
            h, w = self.images.tensor.shape[-2:]
            if self.sem_targets == "evidence":
                oh_all, sp = self._evidence
                assert tuple(sp.shape[-2:]) == (h, w), "superpixel maps must be padded like the images"
                self.pgt_sem_seg = paint_sem_seg_evidence(pg["idx"], offsets, oh_all, sp, self.stuff_cls, pg["scores"],
                                                          self.stuff_cnt, self.num_classes - 1)
            else:
                self.pgt_sem_seg = paint_sem_seg(pg["boxes"], self.stuff_cls, pg["scores"], self.stuff_cnt,
                                                 self.num_classes - 1, h, w)
        else:
            self.pgt_sem_seg = None

        # K refinement branches: labels of branch k come from branch k-1's predictions (k = 0: MIL scores,
        # raw proposals).  Three launches per round (row lse, top-1 per class, IoU match).
        col = 2
        prev_logits = prev_deltas = None
        for k, refinery in enumerate(self.box_refinery):
            if k == 0:
                pg = mine_top1(scores, all_boxes, offsets, things_cls, things_cnt, img_probs)
            else:
                pg = mine_top1(prev_logits, all_boxes, offsets, things_cls, things_cnt, img_probs,
                               lse=row_lse(prev_logits), deltas=prev_deltas)
            lab = match_label(all_boxes, offsets, pg, things_cls, things_cnt, self.num_classes)
            z = outs[col]
            d = outs[col + 1] if refinery.has_reg else None
            col += 2 if refinery.has_reg else 1
            losses.update(refinery.losses((z, d), all_boxes, lab["labels"], lab["boxes"], lab["weights"]))
            prev_logits, prev_deltas = z.detach(), (d.detach() if d is not None else None)
            self.aux["pgt_idx_r%d" % k] = pg["idx"]          # (B, num_classes) padded; valid: [:things_cnt[b]]
            self.aux["labels_r%d" % k] = lab["labels"]       # (R,) int32
        self.aux["things_cnt"] = things_cnt
        self._last_branch = (prev_logits, prev_deltas)
        return losses

    def _mask_prepare(self):
        """Labels of the mask branch and the per-image foreground counts on their way to the host (one small
        asynchronous copy + event: the step's single synchronisation is the wait for it in `_forward_mask`; running
        the semantic head inside that window was measured and did not pay, 21.2-22.5 against 21.4 ms per step)."""
        all_boxes, offsets, things_cls, things_cnt, counts = self._mining
        prev_logits, prev_deltas = self._last_branch
        pg = mine_top1(prev_logits, all_boxes, offsets, things_cls, things_cnt, self.pred_class_img_logits,
                       lse=row_lse(prev_logits), deltas=prev_deltas)
        lab = match_label(all_boxes, offsets, pg, things_cls, things_cnt, self.num_classes)
        if self.mask_targets == "evidence":
            lab["near_rows"], lab["matched_near"] = near_targets(all_boxes, offsets, lab["labels"], self.num_classes, pg,
                                                                 things_cnt, self.mask_mined_top_k)
        with torch.no_grad():
            # the foreground rows, compacted in row order with everything the branch reads of them, and the per-image
            # counts — one launch (layers/mining.py: fg_compact); after the counts have crossed, slices
            sel = fg_compact(lab["labels"], self.num_classes, offsets, all_boxes,
                             lab["matched_near"] if self.mask_targets == "evidence" else lab["matched"])
            host = getattr(self, "_fg_counts_host", None)
            if host is None or host.numel() != len(counts):
                host = self._fg_counts_host = torch.empty(len(counts), dtype=torch.int64).pin_memory()
            host.copy_(sel["counts"], non_blocking=True)
            ready = torch.cuda.Event()
            ready.record()
        self._mask_pending = (pg, lab, sel, host, ready)

    def _forward_mask(self, features, instances):
        all_boxes, offsets, things_cls, things_cnt, counts = self._mining
        pg, lab, sel, host, ready = self._mask_pending
        self._mask_pending = None
        height, width = self.images.tensor.shape[-2:]
        feats = [features[f] for f in self.mask_in_features]
        with torch.no_grad():
            # the head trains on foreground proposals only: a data-dependent count -> the step's one sync
            ready.synchronize()
            per_image = host.tolist()
            n_fg = sum(per_image)
            fg, gt_classes, fg_boxes, img_of = sel["rows"][:n_fg], sel["classes"][:n_fg], sel["boxes"][:n_fg], sel["img"][:n_fg]
            # targets: the matched pseudo-GT rectangle (shrunk by 2 px) cropped to the proposal at 28x28 with
            # ROIAlign(1.0, sampling 0, aligned) and thresholded at 0.5 (structures/masks.py:169-200) — analytic kernel
            side = 2 * self.mask_pooler.output_size[0]
            if self.mask_targets == "evidence":
                # the matched near target's superpixel-evidence mask, cropped to the proposal (never rasterised)
                oh_all, sp = self._evidence
                gt_masks = sp_mask_targets(fg_boxes, sel["matched"][:n_fg], img_of, oh_all, sp, side)
                self.aux["near_rows"], self.aux["matched_near"] = lab["near_rows"], sel["matched"][:n_fg]
            else:
                G = things_cls.shape[1]
                matched = pg["boxes"].reshape(-1, 4)[img_of.to(torch.int64) * G + sel["matched"][:n_fg].to(torch.int64)]
                gt_masks = rect_mask_targets(fg_boxes, matched, side, height, width)
            self.aux["mask_targets"] = gt_masks
        fg_box_lists = [Boxes(b) for b in fg_boxes.split(per_image)]
        mask_features = self.mask_pooler(feats, fg_box_lists)
        self.aux.update(fg_rois=sel["rois"][:n_fg], fg_classes=gt_classes)
        # the pooled features are read by every mask head: one view each, their gradients meet in one map (layers/grad_fan.py)
        views = list(fan_out(mask_features, 1 + len(self.mask_refinery)))
        pred_mask_logits, _ = self.mask_head.layers(views.pop())
        losses = {"loss_mask": mask_rcnn_loss(pred_mask_logits, gt_classes, gt_masks)}
        for k, head in enumerate(self.mask_refinery):
            with torch.no_grad():
                n = pred_mask_logits.size(0)
                sel = pred_mask_logits.detach()[torch.arange(n, device=gt_classes.device), gt_classes]
                if self.mask_targets == "evidence":   # get_pgt_mask: paste the class probability, crop it back
                    gt_masks = paste_crop_targets(sel.sigmoid(), fg_boxes, side, height, width, 0.5)
                else:
                    gt_masks = sel > 0.0                                        # sigmoid > 0.5
                self.aux["mask_targets_r%d" % k] = gt_masks
            pred_mask_logits, _ = head.layers(views.pop())
            losses["loss_mask_r" + str(k)] = mask_rcnn_loss(pred_mask_logits, gt_classes, gt_masks)
        return losses
