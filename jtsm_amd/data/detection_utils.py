"""Proposal files and their transformation — surface of projects/WSL/wsl/data/detection_utils.py:266-345
(transform_proposals_seg) and of the per-image pickle that projects/WSL/tools/proposal_convert.py writes:

    {"boxes": (R, 4) number array, "scores": (R,) or (R, 1), "oh_labels": (R, L) 0/1 — proposal x superpixel
     membership, "superpixels": (H, W) int — superpixel id of every pixel, "indexes": image id,
     "bbox_mode": int, optional (0 = XYXY_ABS, 1 = XYWH_ABS; detectron2/structures/boxes.py:17-45)}
"""
import pickle

import numpy as np
import torch

from ..structures import Boxes, Instances

XYXY_ABS, XYWH_ABS = 0, 1


def write_proposal_file(path, *, boxes, scores, oh_labels, superpixels, image_id, bbox_mode=None):
    d = dict(boxes=np.asarray(boxes), scores=np.asarray(scores), oh_labels=np.asarray(oh_labels),
             superpixels=np.asarray(superpixels), indexes=image_id)
    if bbox_mode is not None:
        d["bbox_mode"] = int(bbox_mode)
    with open(path, "wb") as f:
        pickle.dump(d, f, pickle.HIGHEST_PROTOCOL)


def read_proposal_file(path):
    with open(path, "rb") as f:
        d = pickle.load(f, encoding="latin1")
    for k in ("boxes", "scores", "oh_labels", "superpixels", "indexes"):
        if k not in d:
            raise KeyError("proposal file %s has no '%s' field" % (path, k))
    return d


def _to_xyxy(boxes, mode):
    if mode == XYXY_ABS:
        return boxes
    if mode == XYWH_ABS:
        out = np.array(boxes, copy=True)
        out[:, 2] += out[:, 0]
        out[:, 3] += out[:, 1]
        return out
    raise NotImplementedError("bbox_mode %r: proposal files hold XYXY_ABS or XYWH_ABS boxes" % (mode,))


def unique_boxes(boxes: torch.Tensor, scale=1.0):
    """Indices (ascending) of the first occurrence of every distinct rounded box
    (detectron2/structures/boxes.py:226-238 of the reference tree)."""
    b = boxes.detach().cpu().numpy()
    hashes = np.round(b * scale).dot(np.array([1, 1e3, 1e6, 1e9])).astype(np.int64)
    _, index = np.unique(hashes, return_index=True)
    return np.sort(index)


def transform_proposals_seg(dataset_dict, image_shape, transforms, *, proposal_topk, min_box_size=0):
    """In place: `proposal_file` -> `proposals` (Instances: proposal_boxes, objectness_logits, oh_labels) and
    `superpixels` (H, W) int32, both in the transformed image's frame."""
    if "proposal_file" not in dataset_dict:
        return
    proposals = read_proposal_file(dataset_dict["proposal_file"])
    superpixels, oh_labels = proposals["superpixels"], proposals["oh_labels"]
    scores = np.asarray(proposals["scores"]).reshape(-1)   # (R,) or (R, 1); np.squeeze there breaks for R == 1
    assert proposals["indexes"] == dataset_dict["image_id"], (proposals["indexes"], dataset_dict["image_id"])
    boxes = transforms.apply_box(_to_xyxy(proposals["boxes"], proposals.get("bbox_mode", XYXY_ABS)))
    boxes = Boxes(torch.as_tensor(np.asarray(boxes, dtype=np.float32)))
    objectness_logits = torch.as_tensor(scores.astype("float32"))
    oh_labels = torch.as_tensor(oh_labels.astype("int32"))
    boxes.clip(image_shape)
    keep = torch.as_tensor(unique_boxes(boxes.tensor))
    boxes, objectness_logits, oh_labels = boxes[keep], objectness_logits[keep], oh_labels[keep]
    keep = boxes.nonempty(threshold=min_box_size)
    boxes, objectness_logits, oh_labels = boxes[keep], objectness_logits[keep], oh_labels[keep]
    out = Instances(image_shape)
    out.proposal_boxes = boxes[:proposal_topk]
    out.objectness_logits = objectness_logits[:proposal_topk]
    out.oh_labels = oh_labels[:proposal_topk]
    dataset_dict["proposals"] = out
    superpixels = transforms.apply_segmentation(superpixels.astype("float32"))
    dataset_dict["superpixels"] = torch.as_tensor(np.ascontiguousarray(superpixels.astype("int32")))
