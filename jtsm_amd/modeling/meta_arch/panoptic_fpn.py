"""combine_semantic_and_instance_outputs — surface of detectron2/modeling/meta_arch/panoptic_fpn.py:133-218.

The reference walks the instances on the host with `.item()` after every reduction; here the walk is a chain of
launches whose accept / reject decisions read counters written by earlier launches (csrc/postprocess.hip: pan_*), and
the segment table comes back in one copy at the end."""
import torch

from ...layers.postprocess import panoptic_combine


@torch.no_grad()
def combine_semantic_and_instance_outputs(instance_results, semantic_results, overlap_threshold, stuff_area_limit,
                                          instances_confidence_threshold, num_sem_classes=256):
    """-> panoptic_seg (H, W) int32, segments_info list of dicts (id, isthing, category_id[, score, instance_id |
    area]).  Instances are visited by descending score (equal scores: lower index first)."""
    n = len(instance_results)
    pan, table, tscore = panoptic_combine(instance_results.pred_masks if n else None,
                                          instance_results.scores if n else None,
                                          instance_results.pred_classes if n else None, semantic_results,
                                          num_sem_classes, overlap_threshold, stuff_area_limit,
                                          instances_confidence_threshold)
    rows, scores = table.tolist(), tscore.tolist()
    segments_info = []
    for (sid, isthing, category, inst, area), score in zip(rows, scores):
        if isthing:
            segments_info.append({"id": sid, "isthing": True, "score": score, "category_id": category,
                                  "instance_id": inst})
        else:
            segments_info.append({"id": sid, "isthing": False, "category_id": category, "area": area})
    return pan, segments_info
