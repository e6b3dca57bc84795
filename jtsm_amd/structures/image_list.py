"""ImageList — surface of detectron2/structures/image_list.py:71-125: images of different sizes
padded (bottom/right) into one batch tensor whose H, W are multiples of size_divisibility."""
from typing import List, Tuple

import torch


class ImageList:
    def __init__(self, tensor: torch.Tensor, image_sizes: List[Tuple[int, int]]):
        self.tensor = tensor
        self.image_sizes = image_sizes

    def __len__(self):
        return len(self.image_sizes)

    def __getitem__(self, idx):
        size = self.image_sizes[idx]
        return self.tensor[idx, ..., : size[0], : size[1]]

    def to(self, *args, **kwargs):
        return ImageList(self.tensor.to(*args, **kwargs), self.image_sizes)

    @property
    def device(self):
        return self.tensor.device

    @staticmethod
    def from_tensors(tensors, size_divisibility: int = 0, pad_value: float = 0.0, channels_last: bool = False):
        assert len(tensors) > 0 and isinstance(tensors, (tuple, list))
        for t in tensors:
            assert isinstance(t, torch.Tensor), type(t)
            assert t.shape[:-2] == tensors[0].shape[:-2], t.shape
        image_sizes = [(im.shape[-2], im.shape[-1]) for im in tensors]
        max_h = max(s[0] for s in image_sizes)
        max_w = max(s[1] for s in image_sizes)
        if size_divisibility > 1:
            max_h = (max_h + size_divisibility - 1) // size_divisibility * size_divisibility
            max_w = (max_w + size_divisibility - 1) // size_divisibility * size_divisibility
        batch_shape = [len(tensors)] + list(tensors[0].shape[:-2]) + [max_h, max_w]
        if all(s == (max_h, max_w) for s in image_sizes):
            # nothing to pad (equal sizes, already divisible): one gather instead of a fill and a copy per image
            batched = torch.stack(list(tensors))
            if channels_last and batched.dim() == 4:
                batched = batched.contiguous(memory_format=torch.channels_last)
            return ImageList(batched, image_sizes)
        if channels_last and len(batch_shape) == 4:
            batched = tensors[0].new_full(batch_shape, pad_value).contiguous(memory_format=torch.channels_last)
        else:
            batched = tensors[0].new_full(batch_shape, pad_value)
        for img, pad_img in zip(tensors, batched):
            pad_img[..., : img.shape[-2], : img.shape[-1]].copy_(img)
        return ImageList(batched.contiguous(memory_format=torch.channels_last)
                         if channels_last and batched.dim() == 4 else batched, image_sizes)
