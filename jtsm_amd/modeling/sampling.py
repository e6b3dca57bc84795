"""subsample_labels — detectron2/modeling/sampling.py:9-54: pick up to `num_samples` labelled elements, at most
`positive_fraction` of them positive, uniformly at random (torch.randperm, as the reference)."""
import torch


def subsample_labels(labels: torch.Tensor, num_samples: int, positive_fraction: float, bg_label: int):
    """labels: -1 = ignore, bg_label = negative, anything else positive.  -> (pos_idx, neg_idx)."""
    positive = torch.nonzero((labels != -1) & (labels != bg_label)).squeeze(1)
    negative = torch.nonzero(labels == bg_label).squeeze(1)
    num_pos = min(positive.numel(), int(num_samples * positive_fraction))
    num_neg = min(negative.numel(), num_samples - num_pos)
    pos = positive[torch.randperm(positive.numel(), device=positive.device)[:num_pos]]
    neg = negative[torch.randperm(negative.numel(), device=negative.device)[:num_neg]]
    return pos, neg
