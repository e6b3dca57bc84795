"""Boxes / pairwise_iou — the part of detectron2/structures/boxes.py:143-392 the JTSM path uses: a thin wrapper of an
(N, 4) float32 tensor of absolute (x0, y0, x1, y1) corners.  Method names are the reference's (every caller in
`modeling/` is written against them); the bodies are this repo's."""
import torch


def _as_corners(data):
    """Anything tensor-like -> (N, 4) float32 on its own device; empty input -> (0, 4)."""
    t = data if isinstance(data, torch.Tensor) else torch.as_tensor(data, dtype=torch.float32)
    t = t.to(torch.float32)
    if t.numel() == 0:
        return t.new_zeros((0, 4))
    if t.dim() != 2 or t.shape[1] != 4:
        raise ValueError("Boxes wants an (N, 4) tensor, got %s" % (tuple(t.shape),))
    return t


class Boxes(object):
    __slots__ = ("tensor",)

    def __init__(self, tensor):
        self.tensor = _as_corners(tensor)

    # ---- container behaviour
    def __len__(self):
        return self.tensor.shape[0]

    def __iter__(self):
        return iter(self.tensor)

    def __getitem__(self, index):
        """An int picks one box (still a Boxes of length 1); slices / masks / index tensors pick a subset."""
        rows = self.tensor[index]
        if rows.dim() == 1:
            rows = rows.unsqueeze(0)
        return Boxes(rows)

    def __repr__(self):
        return "Boxes(%s)" % (self.tensor,)

    @property
    def device(self):
        return self.tensor.device

    def to(self, device):
        return Boxes(self.tensor.to(device=device))

    def clone(self):
        return Boxes(self.tensor.clone())

    @classmethod
    def cat(cls, boxes_list):
        rows = [b.tensor for b in boxes_list]
        return cls(torch.cat(rows, dim=0) if rows else torch.empty(0))

    # ---- geometry
    def _wh(self):
        return self.tensor[:, 2:] - self.tensor[:, :2]

    def area(self):
        return self._wh().prod(dim=1)

    def get_centers(self):
        return self.tensor.reshape(-1, 2, 2).mean(dim=1)

    def nonempty(self, threshold=0.0):
        """Mask of the boxes whose width AND height exceed `threshold`."""
        return (self._wh() > threshold).all(dim=1)

    def clip(self, box_size):
        """In place: corners limited to the image [0, w] x [0, h]; `box_size` is (h, w)."""
        h, w = box_size
        limit = self.tensor.new_tensor([w, h, w, h])
        torch.minimum(self.tensor.clamp_(min=0), limit, out=self.tensor)

    def scale(self, scale_x, scale_y):
        """In place: x coordinates times scale_x, y coordinates times scale_y."""
        self.tensor.mul_(self.tensor.new_tensor([scale_x, scale_y, scale_x, scale_y]))


def pairwise_iou(boxes1, boxes2):
    """(N, M) intersection over union; 0 where two boxes do not meet (boxes.py:345-392)."""
    a, b = boxes1.tensor[:, None, :], boxes2.tensor[None, :, :]
    overlap = (torch.minimum(a[..., 2:], b[..., 2:]) - torch.maximum(a[..., :2], b[..., :2])).clamp_(min=0).prod(dim=2)
    union = boxes1.area()[:, None] + boxes2.area()[None, :] - overlap
    return torch.where(overlap > 0, overlap / union, overlap.new_zeros(()))
