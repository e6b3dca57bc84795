"""Box2BoxTransform — call surface of detectron2/modeling/box_regression.py:21-113 (R-CNN box deltas
(dx, dy, dw, dh) with per-coordinate weights), written on top of two small xyxy <-> centre/size helpers."""
import math

import torch

_DEFAULT_SCALE_CLAMP = math.log(1000.0 / 16)


def _to_center_size(boxes):
    """(..., 4) xyxy -> centre (..., 2) and size (..., 2)."""
    size = boxes[..., 2:] - boxes[..., :2]
    return boxes[..., :2] + 0.5 * size, size


class Box2BoxTransform:
    def __init__(self, weights, scale_clamp: float = _DEFAULT_SCALE_CLAMP):
        self.weights = weights          # (wx, wy, ww, wh)
        self.scale_clamp = scale_clamp  # upper bound of dw, dh before exp()

    def get_deltas(self, src_boxes, target_boxes):
        """Deltas that map src_boxes (N,4) onto target_boxes (N,4)."""
        assert isinstance(src_boxes, torch.Tensor) and isinstance(target_boxes, torch.Tensor)
        sc, ss = _to_center_size(src_boxes)
        tc, ts = _to_center_size(target_boxes)
        assert bool((ss[:, 0] > 0).all()), "Input boxes to Box2BoxTransform are not valid!"
        w = src_boxes.new_tensor(self.weights)
        shift = w[:2] * (tc - sc) / ss
        scale = w[2:] * torch.log(ts / ss)
        return torch.cat((shift, scale), dim=1)

    def apply_deltas(self, deltas, boxes):
        """deltas (N, k*4) for k classes, boxes (N,4) -> predicted boxes (N, k*4)."""
        boxes = boxes.to(deltas.dtype)
        n = deltas.shape[0]
        d = deltas.reshape(n, -1, 4)
        centre, size = _to_center_size(boxes)
        centre, size = centre[:, None, :], size[:, None, :]
        w = deltas.new_tensor(self.weights)
        new_centre = d[..., :2] / w[:2] * size + centre
        new_size = torch.exp(torch.clamp(d[..., 2:] / w[2:], max=self.scale_clamp)) * size
        out = torch.cat((new_centre - 0.5 * new_size, new_centre + 0.5 * new_size), dim=-1)
        return out.reshape(n, -1)
