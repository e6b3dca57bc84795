"""Test-time augmentation with averaging — surface of projects/WSL/wsl/modeling/test_time_augmentation_avg.py:
DatasetMapperTTAAVG (:125-196) and GeneralizedRCNNWithTTAAVG (:199-442).

Every augmented view (shortest-edge resize, optionally mirrored) is run through the model's raw inference; class
probabilities and the boxes (mapped back to the original frame) are AVERAGED over the views, one detection pass runs
on the averages, the masks of the merged detections are predicted in every view and averaged, and so are the
semantic logits.  The reference moves boxes and score maps to the host for every inverse transform
(`.cpu().numpy()`, :371-381, :436-440); here the inverse transforms are a few device operations
(scale / mirror of box columns, `jtsm_resize_nearest_f32` with mirrored source columns)."""
import copy
from contextlib import contextmanager

import numpy as np
import torch
from torch import nn

from ..data.transforms import (HFlipTransform, NoOpTransform, RandomFlip, ResizeShortestEdge, ResizeTransform,
                               TransformList, apply_augmentations)
from ..layers.postprocess import resize_nearest
from ..structures import Boxes, Instances
from .meta_arch import GeneralizedMCNNWSL
from .roi_heads.fast_rcnn_oicr import fast_rcnn_inference_single_image

__all__ = ["DatasetMapperTTAAVG", "GeneralizedRCNNWithTTAAVG"]


def _box_map(tfm: TransformList):
    """The (resize, flip) chain as x' = ax * x + bx (mirrored or not), y' = ay * y — enough for every transform the
    TTA mapper builds; used to move boxes on the device."""
    ax, bx, ay, mirrored = 1.0, 0.0, 1.0, False
    for t in tfm.transforms:
        if isinstance(t, NoOpTransform):
            continue
        if isinstance(t, ResizeTransform):
            sx, sy = t.new_w * 1.0 / t.w, t.new_h * 1.0 / t.h
            ax, bx, ay = ax * sx, bx * sx, ay * sy
        elif isinstance(t, HFlipTransform):
            ax, bx, mirrored = -ax, t.width - bx, not mirrored
        else:
            raise NotImplementedError("TTA handles resize and horizontal flip (%s)" % type(t).__name__)
    return ax, bx, ay, mirrored


def apply_box_device(tfm: TransformList, boxes: torch.Tensor):
    """Transform.apply_box for (N, 4) device boxes: corners through the affine map, then min / max."""
    ax, bx, ay, _ = _box_map(tfm)
    x0, x1 = boxes[:, 0] * ax + bx, boxes[:, 2] * ax + bx
    return torch.stack([torch.minimum(x0, x1), boxes[:, 1] * ay, torch.maximum(x0, x1), boxes[:, 3] * ay], dim=1)


def transform_proposals(dataset_dict, image_shape, transforms, *, proposal_topk, min_box_size=0):
    """test_time_augmentation_avg.py:28-122 for an input that already carries `proposals` / `superpixels`."""
    prop = dataset_dict["proposals"]
    boxes = Boxes(torch.as_tensor(np.asarray(transforms.apply_box(prop.proposal_boxes.tensor.cpu().numpy()),
                                             dtype=np.float32)))
    boxes.clip(image_shape)
    keep = boxes.nonempty(threshold=min_box_size)
    out = Instances(image_shape)
    out.proposal_boxes = boxes[keep][:proposal_topk]
    out.objectness_logits = prop.objectness_logits.cpu()[keep][:proposal_topk]
    if prop.has("oh_labels"):
        out.oh_labels = prop.oh_labels.cpu()[keep][:proposal_topk]
    dataset_dict["proposals"] = out
    if "superpixels" in dataset_dict:
        sp = transforms.apply_segmentation(dataset_dict["superpixels"].cpu().numpy().astype("float32"))
        dataset_dict["superpixels"] = torch.as_tensor(np.ascontiguousarray(sp.astype("int32")))


class DatasetMapperTTAAVG:
    def __init__(self, cfg):
        self.min_sizes = cfg.TEST.AUG.MIN_SIZES
        self.max_size = cfg.TEST.AUG.MAX_SIZE
        self.flip = cfg.TEST.AUG.FLIP
        self.image_format = cfg.INPUT.FORMAT
        self.proposal_topk = cfg.DATASETS.PRECOMPUTED_PROPOSAL_TOPK_TEST if cfg.MODEL.LOAD_PROPOSALS else None

    def __call__(self, dataset_dict):
        numpy_image = dataset_dict["image"].permute(1, 2, 0).cpu().numpy()
        shape = numpy_image.shape
        orig_shape = (dataset_dict["height"], dataset_dict["width"])
        pre_tfm = ResizeTransform(orig_shape[0], orig_shape[1], shape[0], shape[1]) if shape[:2] != orig_shape \
            else NoOpTransform()
        aug_candidates = []
        for min_size in self.min_sizes:
            resize = ResizeShortestEdge(min_size, self.max_size)
            aug_candidates.append([resize])
            if self.flip:
                aug_candidates.append([resize, RandomFlip(prob=1.0)])
        ret = []
        for aug in aug_candidates:
            new_image, tfms = apply_augmentations(aug, np.copy(numpy_image))
            dic = copy.deepcopy(dataset_dict)
            dic["transforms"] = TransformList([pre_tfm]) + tfms
            dic["image"] = torch.from_numpy(np.ascontiguousarray(new_image.transpose(2, 0, 1)))
            if self.proposal_topk is not None:
                transform_proposals(dic, new_image.shape[:2], tfms, proposal_topk=self.proposal_topk)
            ret.append(dic)
        return ret


class GeneralizedRCNNWithTTAAVG(nn.Module):
    def __init__(self, cfg, model, tta_mapper=None, batch_size=1):
        super().__init__()
        if isinstance(model, nn.parallel.DistributedDataParallel):
            model = model.module
        assert isinstance(model, GeneralizedMCNNWSL), \
            "TTA is only supported on GeneralizedMCNNWSL. Got a model of type {}".format(type(model))
        self.cfg = cfg.clone() if hasattr(cfg, "clone") else cfg
        self.model = model
        self.tta_mapper = DatasetMapperTTAAVG(cfg) if tta_mapper is None else tta_mapper
        self.batch_size = batch_size
        self.is_ps = True

    @contextmanager
    def _turn_off_roi_heads(self, attrs):
        roi_heads = self.model.roi_heads
        old = {a: getattr(roi_heads, a) for a in attrs if hasattr(roi_heads, a)}
        for a in old:
            setattr(roi_heads, a, False)
        try:
            yield
        finally:
            for a, v in old.items():
                setattr(roi_heads, a, v)

    def _batch_inference(self, batched_inputs, detected_instances=None, only_sem_seg=False):
        if detected_instances is None:
            detected_instances = [None] * len(batched_inputs)
        outputs, all_scores, all_boxes = [], [], []
        inputs, instances = [], []
        for idx, (inp, inst) in enumerate(zip(batched_inputs, detected_instances)):
            inputs.append(inp)
            instances.append(inst)
            if len(inputs) == self.batch_size or idx == len(batched_inputs) - 1:
                output, score, box = self.model.inference(inputs, instances if instances[0] is not None else None,
                                                          do_postprocess=False, only_sem_seg=only_sem_seg)
                outputs.extend(output)
                if score is not None and len(score):
                    all_scores.extend(score)
                    all_boxes.extend(box)
                inputs, instances = [], []
        return outputs, all_scores, all_boxes

    @torch.no_grad()
    def __call__(self, batched_inputs):
        def _prepare(d):
            ret = copy.copy(d)
            if "height" not in ret and "width" not in ret:
                ret["height"], ret["width"] = ret["image"].shape[1], ret["image"].shape[2]
            return ret

        return [self._inference_one_image(_prepare(x)) for x in batched_inputs]

    def _inference_one_image(self, input):
        orig_shape = (input["height"], input["width"])
        augmented_inputs = self.tta_mapper(input)
        tfms = [x.pop("transforms") for x in augmented_inputs]
        with self._turn_off_roi_heads(["mask_on", "keypoint_on"]):
            all_boxes, all_scores = self._get_augmented_boxes(augmented_inputs, tfms)
        merged_instances = self._merge_detections(all_boxes, all_scores, orig_shape)
        if not self.cfg.MODEL.MASK_ON:
            return {"instances": merged_instances}
        augmented_instances = self._rescale_detected_boxes(augmented_inputs, merged_instances, tfms)
        outputs, _, _ = self._batch_inference(augmented_inputs, augmented_instances)
        merged_instances.pred_masks = self._reduce_pred_masks(outputs, tfms)
        outputs, _, _ = self._batch_inference(augmented_inputs, only_sem_seg=True)
        sem_seg = self._reduce_pred_sem_seg(augmented_inputs, outputs, tfms, orig_shape)
        return self.model._postprocess_ps([sem_seg], [merged_instances], [input], [orig_shape])[0]

    def _get_augmented_boxes(self, augmented_inputs, tfms):
        _, all_scores, all_boxes = self._batch_inference(augmented_inputs)
        back = []
        for pred_boxes, tfm in zip(all_boxes, tfms):
            num_img, num_pred, num_col = pred_boxes.shape
            assert num_img == 1
            back.append(apply_box_device(tfm.inverse(), pred_boxes.reshape(num_pred * num_col // 4, 4))
                        .reshape(1, num_pred, num_col))
        boxes = torch.mean(torch.cat(back, dim=0), dim=0)
        scores = torch.mean(torch.cat(all_scores, dim=0), dim=0)
        return boxes, scores

    def _merge_detections(self, all_boxes, all_scores, shape_hw):
        merged, _, _, _ = fast_rcnn_inference_single_image(
            all_boxes, all_scores, shape_hw, self.cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST,
            self.cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST, self.cfg.TEST.DETECTIONS_PER_IMAGE)
        return merged

    def _rescale_detected_boxes(self, augmented_inputs, merged_instances, tfms):
        out = []
        for inp, tfm in zip(augmented_inputs, tfms):
            out.append(Instances(image_size=tuple(inp["image"].shape[1:3]),
                                 pred_boxes=Boxes(apply_box_device(tfm, merged_instances.pred_boxes.tensor)),
                                 pred_classes=merged_instances.pred_classes, scores=merged_instances.scores))
        return out

    def _reduce_pred_masks(self, outputs, tfms):
        masks = [o.pred_masks.flip(dims=[3]) if _box_map(t)[3] else o.pred_masks for o, t in zip(outputs, tfms)]
        return torch.mean(torch.stack(masks, dim=0), dim=0)

    def _reduce_pred_sem_seg(self, inputs, outputs, tfms, orig_shape):
        all_sem_seg = []
        for inp, output, tfm in zip(inputs, outputs, tfms):
            h, w = inp["image"].shape[1:3]
            at_view = resize_nearest(output, (h, w))                      # F.interpolate(..., size=(h, w), "nearest")
            # inverse transforms: un-mirror, then nearest resize back to the original frame
            all_sem_seg.append(resize_nearest(at_view, orig_shape, flip_source=_box_map(tfm)[3]))
        return torch.mean(torch.stack(all_sem_seg, dim=0), dim=0)
