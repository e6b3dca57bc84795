"""Shared by the model parity tests, bench.py and __graft_entry__.smoke(): turn oracle/model.py's
synthetic batch (SURVEY §8d recipe) into the reference's list[dict] input contract."""
import os

import torch

from jtsm_amd.config import get_cfg
from jtsm_amd.structures import Boxes, Instances

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def jtsm_cfg(device="cuda", depth=50):
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "jtsm_R_50_FPN_1x.yaml"))
    cfg.MODEL.DEVICE = device
    cfg.MODEL.RESNETS.DEPTH = depth
    return cfg


def to_batched_inputs(batch):
    out = []
    for i, img in enumerate(batch["images"]):
        size = tuple(img.shape[-2:])
        out.append({
            "image": img,
            "instances": Instances(size, gt_classes=batch["gt_classes"][i]),
            "sem_seg": batch["sem_seg"][i],
            "superpixels": batch["superpixels"][i],
            "proposals": Instances(size, proposal_boxes=Boxes(batch["boxes"][i]),
                                   objectness_logits=batch["objectness"][i], oh_labels=batch["oh_labels"][i]),
        })
    return out
