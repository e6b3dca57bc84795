"""Matcher — same call surface as detectron2/modeling/matcher.py:8-126 (thresholds/labels in, per-prediction
(matched gt index, label in {-1,0,1}) out), implemented as one bucketisation of the best IoU per prediction
instead of a loop of range masks: label = labels[#thresholds <= best_iou]."""
from typing import List

import torch


class Matcher:
    def __init__(self, thresholds: List[float], labels: List[int], allow_low_quality_matches: bool = False):
        if not thresholds or thresholds[0] <= 0:
            raise ValueError("thresholds must be positive")
        if any(a > b for a, b in zip(thresholds[:-1], thresholds[1:])):
            raise ValueError("thresholds must be non-decreasing")
        if len(labels) != len(thresholds) + 1 or any(l not in (-1, 0, 1) for l in labels):
            raise ValueError("need len(thresholds)+1 labels, each in {-1, 0, 1}")
        self.thresholds = [-float("inf")] + list(thresholds) + [float("inf")]   # kept for introspection
        self.labels = list(labels)
        self.allow_low_quality_matches = allow_low_quality_matches
        self._cuts = torch.tensor(list(thresholds), dtype=torch.float32)
        self._label_table = torch.tensor(list(labels), dtype=torch.int8)

    def __call__(self, match_quality_matrix: torch.Tensor):
        """match_quality_matrix: (num_gt, num_predictions), entries >= 0.  Returns (matches int64 (N,),
        match_labels int8 (N,)).  With no ground truth every prediction gets index 0 and labels[0]."""
        q = match_quality_matrix
        assert q.dim() == 2
        n = q.size(1)
        if q.numel() == 0:
            return (q.new_zeros((n,), dtype=torch.int64),
                    q.new_full((n,), self.labels[0], dtype=torch.int8))
        assert bool(torch.all(q >= 0))
        best, which = q.max(dim=0)
        cuts = self._cuts.to(device=q.device, dtype=best.dtype)
        # number of thresholds t with t <= best  ==  index of the [low, high) interval containing best
        interval = torch.bucketize(best, cuts, right=True)
        out = self._label_table.to(q.device)[interval]
        if self.allow_low_quality_matches:
            # every ground truth keeps its best prediction(s), whatever their IoU
            top_per_gt = q.max(dim=1, keepdim=True).values
            out[(q == top_per_gt).any(dim=0)] = 1
        return which, out
