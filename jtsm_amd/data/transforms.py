"""Geometric transforms of the mappers — surface of detectron2/data/transforms/transform.py:94-158 (ResizeTransform),
fvcore.transforms.HFlipTransform / NoOpTransform / TransformList (fvcore >= 0.1.2, setup.py:214; absent from
/root/reference: restated from their published behaviour: flip = reverse the width axis, x -> w - x; boxes go through
their four corners and come back as the min / max), and detectron2/data/transforms/augmentation_impl.py:67-172
(RandomFlip, ResizeShortestEdge)."""
import sys

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image


class Transform:
    def apply_image(self, img, interp=None):
        raise NotImplementedError

    def apply_coords(self, coords):
        raise NotImplementedError

    def apply_segmentation(self, segmentation):
        return self.apply_image(segmentation)

    def apply_box(self, box):
        """(N, 4) XYXY -> the axis-aligned hull of the transformed corners."""
        idxs = np.array([(0, 1), (2, 1), (0, 3), (2, 3)]).flatten()
        coords = np.asarray(box).reshape(-1, 4)[:, idxs].reshape(-1, 2)
        coords = self.apply_coords(coords).reshape((-1, 4, 2))
        minxy, maxxy = coords.min(axis=1), coords.max(axis=1)
        return np.concatenate((minxy, maxxy), axis=1)

    def inverse(self):
        raise NotImplementedError


class NoOpTransform(Transform):
    def apply_image(self, img, interp=None):
        return img

    def apply_coords(self, coords):
        return coords

    def inverse(self):
        return self


class HFlipTransform(Transform):
    def __init__(self, width: int):
        self.width = width

    def apply_image(self, img, interp=None):
        return np.flip(img, axis=1) if img.ndim <= 3 else np.flip(img, axis=-2)

    def apply_coords(self, coords):
        coords[:, 0] = self.width - coords[:, 0]
        return coords

    def inverse(self):
        return self


class ResizeTransform(Transform):
    """uint8 images are resized by PIL (bilinear by default), anything else by F.interpolate — as the reference."""

    def __init__(self, h, w, new_h, new_w, interp=None):
        self.h, self.w, self.new_h, self.new_w = h, w, new_h, new_w
        self.interp = Image.BILINEAR if interp is None else interp

    def apply_image(self, img, interp=None):
        assert img.shape[:2] == (self.h, self.w), (img.shape, self.h, self.w)
        interp_method = interp if interp is not None else self.interp
        if img.dtype == np.uint8:
            one = len(img.shape) > 2 and img.shape[2] == 1
            pil = Image.fromarray(img[:, :, 0], mode="L") if one else Image.fromarray(img)
            ret = np.asarray(pil.resize((self.new_w, self.new_h), interp_method))
            return np.expand_dims(ret, -1) if one else ret
        if any(x < 0 for x in img.strides):
            img = np.ascontiguousarray(img)
        t = torch.from_numpy(img)
        shape = list(t.shape)
        shape_4d = shape[:2] + [1] * (4 - len(shape)) + shape[2:]
        t = t.view(shape_4d).permute(2, 3, 0, 1)
        mode = {Image.NEAREST: "nearest", Image.BILINEAR: "bilinear", Image.BICUBIC: "bicubic"}[interp_method]
        t = F.interpolate(t, (self.new_h, self.new_w), mode=mode, align_corners=None if mode == "nearest" else False)
        shape[:2] = (self.new_h, self.new_w)
        return t.permute(2, 3, 0, 1).reshape(shape).numpy()

    def apply_coords(self, coords):
        coords[:, 0] = coords[:, 0] * (self.new_w * 1.0 / self.w)
        coords[:, 1] = coords[:, 1] * (self.new_h * 1.0 / self.h)
        return coords

    def apply_segmentation(self, segmentation):
        return self.apply_image(segmentation, interp=Image.NEAREST)

    def inverse(self):
        return ResizeTransform(self.new_h, self.new_w, self.h, self.w, self.interp)


class TransformList(Transform):
    def __init__(self, transforms):
        self.transforms = []
        for t in transforms:
            self.transforms.extend(t.transforms if isinstance(t, TransformList) else [t])

    def apply_image(self, img, interp=None):
        for t in self.transforms:
            img = t.apply_image(img)
        return img

    def apply_coords(self, coords):
        for t in self.transforms:
            coords = t.apply_coords(coords)
        return coords

    def apply_box(self, box):
        for t in self.transforms:
            box = t.apply_box(box)
        return box

    def apply_segmentation(self, segmentation):
        for t in self.transforms:
            segmentation = t.apply_segmentation(segmentation)
        return segmentation

    def inverse(self):
        return TransformList([t.inverse() for t in self.transforms[::-1]])

    def __add__(self, other):
        return TransformList(self.transforms + (other.transforms if isinstance(other, TransformList) else [other]))

    def __radd__(self, other):
        return TransformList((other.transforms if isinstance(other, TransformList) else [other]) + self.transforms)

    def __len__(self):
        return len(self.transforms)

    def __iter__(self):
        return iter(self.transforms)


NoOpTransform.__add__ = lambda self, other: TransformList([self]) + other


class ResizeShortestEdge:
    def __init__(self, short_edge_length, max_size=sys.maxsize, sample_style="range", interp=Image.BILINEAR, rng=None):
        assert sample_style in ["range", "choice"], sample_style
        self.is_range = sample_style == "range"
        if isinstance(short_edge_length, int):
            short_edge_length = (short_edge_length, short_edge_length)
        self.short_edge_length, self.max_size, self.interp = short_edge_length, max_size, interp
        self.rng = rng if rng is not None else np.random

    def get_transform(self, image):
        h, w = image.shape[:2]
        if self.is_range:
            size = self.rng.randint(self.short_edge_length[0], self.short_edge_length[1] + 1)
        else:
            size = self.rng.choice(self.short_edge_length)
        if size == 0:
            return NoOpTransform()
        scale = size * 1.0 / min(h, w)
        newh, neww = (size, scale * w) if h < w else (scale * h, size)
        if max(newh, neww) > self.max_size:
            scale = self.max_size * 1.0 / max(newh, neww)
            newh, neww = newh * scale, neww * scale
        return ResizeTransform(h, w, int(newh + 0.5), int(neww + 0.5), self.interp)


class RandomFlip:
    def __init__(self, prob=0.5, *, horizontal=True, vertical=False, rng=None):
        if vertical or not horizontal:
            raise NotImplementedError("the JTSM mappers flip horizontally only")
        self.prob = prob
        self.rng = rng if rng is not None else np.random

    def get_transform(self, image):
        return HFlipTransform(image.shape[1]) if self.rng.uniform() < self.prob else NoOpTransform()


def apply_augmentations(augmentations, image):
    """-> (transformed image, TransformList), each augmentation seeing the image left by the previous one."""
    tfms = []
    for aug in augmentations:
        t = aug.get_transform(image) if hasattr(aug, "get_transform") else aug
        image = t.apply_image(image)
        tfms.append(t)
    return image, TransformList(tfms)
