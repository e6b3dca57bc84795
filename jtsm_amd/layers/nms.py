"""batched_nms — surface of detectron2/layers/nms.py:10-31 (which defers to torchvision.ops.boxes.batched_nms below
40000 boxes and loops over classes above).  One library call (csrc/postprocess.hip): sort by (class, score), 64x64 IoU
bit matrix per class, one wavefront per class resolves its chain, second sort orders the survivors.

Ordering contract: survivors by descending score; equal scores keep ascending index (the reference's sort is
unstable there, so any order is "the reference's")."""
import torch

from .postprocess import batched_nms_device


@torch.no_grad()
def batched_nms(boxes: torch.Tensor, scores: torch.Tensor, idxs: torch.Tensor, iou_threshold: float):
    assert boxes.shape[-1] == 4
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    idxs = idxs.to(torch.int64)
    if int(idxs.min()) < 0:
        raise RuntimeError("batched_nms: class indices must be non-negative")
    per_class = torch.bincount(idxs)                       # the op's one sizing read (the result length is
    num_classes, biggest = per_class.numel(), int(per_class.max())   # data-dependent anyway)
    keep, num, ovf = batched_nms_device(boxes, scores, idxs, iou_threshold, num_classes, biggest)
    n, o = torch.cat([num, ovf]).tolist()
    if o:
        raise RuntimeError("batched_nms: internal class-size bound exceeded")
    return keep[:n]
