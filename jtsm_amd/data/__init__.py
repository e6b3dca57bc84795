"""Input pipeline -> device contract of the JTSM path: the proposal file format, the geometric
transforms the mappers apply, the WSL dataset mapper, and a pinned-memory prefetcher that hands the model batches
already resident in HBM.  Host-side code, like the reference's `detectron2.data` / `wsl.data`."""
from .dataset_mapper import DatasetMapper
from .detection_utils import (read_proposal_file, transform_proposals_seg, unique_boxes, write_proposal_file)
from .prefetch import DevicePrefetcher
from .transforms import (HFlipTransform, NoOpTransform, RandomFlip, ResizeShortestEdge, ResizeTransform, TransformList,
                         apply_augmentations)

__all__ = [k for k in globals().keys() if not k.startswith("_")]
