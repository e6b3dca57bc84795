"""RotatedBoxes — the part of detectron2/structures/rotated_boxes.py:14-230 a pooler needs: a thin wrapper of an
(N, 5) float32 tensor of (x_center, y_center, width, height, angle in degrees, counter-clockwise positive).  Method
names are the reference's; the bodies are this repo's.  Rotated IoU / NMS stay out of scope."""
import torch


class RotatedBoxes(object):
    __slots__ = ("tensor",)

    def __init__(self, tensor):
        t = tensor if isinstance(tensor, torch.Tensor) else torch.as_tensor(tensor, dtype=torch.float32)
        t = t.to(torch.float32)
        if t.numel() == 0:
            t = t.new_zeros((0, 5))
        if t.dim() != 2 or t.shape[1] != 5:
            raise ValueError("RotatedBoxes wants an (N, 5) tensor, got %s" % (tuple(t.shape),))
        self.tensor = t

    def __len__(self):
        return self.tensor.shape[0]

    def __iter__(self):
        return iter(self.tensor)

    def __getitem__(self, index):
        rows = self.tensor[index]
        if rows.dim() == 1:
            rows = rows.unsqueeze(0)
        return RotatedBoxes(rows)

    def __repr__(self):
        return "RotatedBoxes(%s)" % (self.tensor,)

    @property
    def device(self):
        return self.tensor.device

    def to(self, device):
        return RotatedBoxes(self.tensor.to(device=device))

    def clone(self):
        return RotatedBoxes(self.tensor.clone())

    @classmethod
    def cat(cls, boxes_list):
        rows = [b.tensor for b in boxes_list]
        return cls(torch.cat(rows, dim=0) if rows else torch.empty(0))

    def area(self):
        """width x height: what assigns a rotated box its FPN level (rotated_boxes.py:238-246)."""
        return self.tensor[:, 2] * self.tensor[:, 3]

    def get_centers(self):
        return self.tensor[:, :2]

    def nonempty(self, threshold=0.0):
        return (self.tensor[:, 2] > threshold) & (self.tensor[:, 3] > threshold)

    def scale(self, scale_x, scale_y):
        """In place, for scale_x == scale_y (the resize the JTSM path applies); an anisotropic resize changes the
        angle too (rotated_boxes.py:390-470) and is refused here."""
        if scale_x != scale_y:
            raise NotImplementedError("anisotropic scaling of rotated boxes is outside the JTSM path")
        self.tensor[:, :4] *= scale_x
