"""DatasetMapper — surface of projects/WSL/wsl/data/dataset_mapper.py:20-187 for the JTSM path: read the image,
apply the augmentations to image / semantic map / proposals / superpixels, emit the model's input dict
(`image` uint8 (3, H, W), `sem_seg` long, `proposals`, `superpixels`, `instances` with image-level `gt_classes`).
Box / mask / keypoint annotations are not used by the weakly supervised heads (only their classes are)."""
import copy

import numpy as np
import torch
from PIL import Image

from ..structures import Instances
from . import detection_utils as utils
from .transforms import RandomFlip, ResizeShortestEdge, apply_augmentations


def read_image(file_name, format="BGR"):
    """detectron2/data/detection_utils.py:166-190 for the two formats the configs use."""
    with Image.open(file_name) as image:
        if format == "L":
            return np.expand_dims(np.asarray(image.convert("L")), -1)
        image = np.asarray(image.convert("RGB"))
    return image[:, :, ::-1] if format == "BGR" else image


class DatasetMapper:
    def __init__(self, is_train: bool, *, augmentations, image_format="BGR", precomputed_proposal_topk=None):
        self.is_train = is_train
        self.augmentations = list(augmentations)
        self.image_format = image_format
        self.proposal_topk = precomputed_proposal_topk

    @classmethod
    def from_config(cls, cfg, is_train: bool = True, rng=None):
        """build_augmentation (wsl/data/detection_utils.py): ResizeShortestEdge (+ RandomFlip in training)."""
        inp = cfg.INPUT
        if is_train:
            augs = [ResizeShortestEdge(inp.MIN_SIZE_TRAIN, inp.MAX_SIZE_TRAIN, inp.MIN_SIZE_TRAIN_SAMPLING, rng=rng),
                    RandomFlip(rng=rng)]
        else:
            augs = [ResizeShortestEdge(inp.MIN_SIZE_TEST, inp.MAX_SIZE_TEST, "choice", rng=rng)]
        topk = None
        if cfg.MODEL.LOAD_PROPOSALS:
            topk = (cfg.DATASETS.PRECOMPUTED_PROPOSAL_TOPK_TRAIN if is_train
                    else cfg.DATASETS.PRECOMPUTED_PROPOSAL_TOPK_TEST)
        return cls(is_train, augmentations=augs, image_format=inp.FORMAT, precomputed_proposal_topk=topk)

    def __call__(self, dataset_dict):
        dataset_dict = copy.deepcopy(dataset_dict)
        image = dataset_dict.pop("image_array") if "image_array" in dataset_dict else \
            read_image(dataset_dict["file_name"], format=self.image_format)
        if "width" in dataset_dict and (image.shape[1], image.shape[0]) != (dataset_dict["width"], dataset_dict["height"]):
            raise ValueError("Mismatched image shape for %s" % dataset_dict.get("file_name", "<array>"))
        if "sem_seg_file_name" in dataset_dict:
            sem_seg_gt = read_image(dataset_dict.pop("sem_seg_file_name"), "L").squeeze(2)
        else:
            sem_seg_gt = dataset_dict.pop("sem_seg_array", None)
        image, transforms = apply_augmentations(self.augmentations, image)
        if sem_seg_gt is not None:
            sem_seg_gt = transforms.apply_segmentation(sem_seg_gt)
        image_shape = image.shape[:2]
        dataset_dict["image"] = torch.as_tensor(np.ascontiguousarray(image.transpose(2, 0, 1)))
        if sem_seg_gt is not None:
            dataset_dict["sem_seg"] = torch.as_tensor(np.ascontiguousarray(sem_seg_gt).astype("long"))
        if self.proposal_topk is not None:
            utils.transform_proposals_seg(dataset_dict, image_shape, transforms, proposal_topk=self.proposal_topk)
        annos = dataset_dict.pop("annotations", None)
        if not self.is_train:
            dataset_dict.pop("sem_seg_file_name", None)
            return dataset_dict
        if annos is not None:   # weak supervision: the image-level class set is all the heads read
            classes = [int(a["category_id"]) for a in annos if a.get("iscrowd", 0) == 0]
            dataset_dict["instances"] = Instances(image_shape, gt_classes=torch.tensor(classes, dtype=torch.int64))
        return dataset_dict
