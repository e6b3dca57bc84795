"""ROIPooler — surface of detectron2/modeling/poolers.py:22-249 and of the WSL variant that adds
MOIPool and the (oh_labels_list, superpixels) arguments (projects/WSL/wsl/modeling/poolers.py:119-331).

MI355X mapping: the reference finds each level's boxes with `nonzero` (a device->host sync per level),
gathers them, pools, and scatters the result back.  Here level assignment stays on the device and each
level is ONE launch over all M boxes that serves only its own (kernel-side filter), writing straight
into the shared (M,C,P,P) output — no sync, no gather/scatter copies.

Multi-level MOIPool is DEFINED here (the reference's multi-level branch is broken): each
box is pooled on its assigned level with that level's scale, and both `output` and `argmax` are
returned for all boxes.
"""
import math
from typing import List

import ctypes as C

import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib as L
from ..layers import grad_fan
from ..layers.mining import MAX_IMAGES, pooler_rois_levels
from ..layers.moi_pool import MOIPool
from ..layers.roi_align import ROIAlign
from ..layers.roi_align_rotated import ROIAlignRotated
from ..layers.wrappers import cat

CL = torch.channels_last


def assign_boxes_to_levels(box_lists, min_level, max_level, canonical_box_size, canonical_level):
    """floor(canonical_level + log2(sqrt(area) / canonical_box_size + 1e-8)) clamped to
    [min_level, max_level], returned 0-based (poolers.py:22-58)."""
    box_sizes = torch.sqrt(cat([boxes.area() for boxes in box_lists]))
    level_assignments = torch.floor(canonical_level + torch.log2(box_sizes / canonical_box_size + 1e-8))
    # (a degenerate box — negative or NaN area — must still land on a level: the level kernels write only the rois
    # they own, and an unowned roi would return uninitialised memory)
    level_assignments = torch.clamp(torch.nan_to_num(level_assignments, nan=float(min_level)), min=min_level,
                                    max=max_level)
    return level_assignments.to(torch.int64) - min_level


def convert_boxes_to_pooler_format(box_lists):
    """list[Boxes] -> (M,5) [batch index, x0, y0, x1, y1]; list[RotatedBoxes] -> (M,6) [batch index, x_ctr, y_ctr,
    width, height, angle] (poolers.py:61-95)."""
    def fmt_box_list(box_tensor, batch_index):
        repeated_index = torch.full((len(box_tensor), 1), batch_index, dtype=box_tensor.dtype,
                                    device=box_tensor.device)
        return cat((repeated_index, box_tensor), dim=1)

    return cat([fmt_box_list(box_list.tensor, i) for i, box_list in enumerate(box_lists)], dim=0)


def _feat(x):
    L.require_gpu(x)
    if x.dtype != torch.float32:
        raise RuntimeError("jtsm_amd multi-level pooling is float32, got %s" % x.dtype)
    return x.contiguous(memory_format=CL)


class _AlignLevels(Function):
    @staticmethod
    def forward(ctx, rois, roi_level, res, sampling_ratio, aligned, scales, fans, *feats):
        """aligned: False / True = ROIAlign / ROIAlignV2 on (M,5) rois; "rotated" = ROIAlignRotated on (M,6) rois.
        fans: the feature levels' fan records (layers/grad_fan.py), claimed by ROIPooler.forward on the caller's tensors."""
        ctx.fans = list(fans) if fans is not None else [None] * len(feats)
        rotated = aligned == "rotated"
        if rois.shape[1] != (6 if rotated else 5):
            raise RuntimeError("pooler rois must be (M, %d), got %s" % (6 if rotated else 5, tuple(rois.shape)))
        feats = [_feat(f) for f in feats]
        M, C = rois.shape[0], feats[0].shape[1]
        out = torch.empty((M, C, res, res), dtype=torch.float32, device=rois.device, memory_format=CL)  # every roi is on exactly one level, whose kernel writes all of its bins
        for lvl, (f, sc) in enumerate(zip(feats, scales)):
            B, _, H, W = f.shape
            # algorithmic bytes: this level's share of the output + the level's map read once (an upper
            # bound on the unique feature bytes touched) + the rois
            L.note_bytes(4.0 * (out.numel() / len(feats) + f.numel() + rois.numel()))
            if rotated:
                L.check(L.lib().jtsm_roi_align_rotated_forward_level_f32(
                    L.ptr(f), L.ptr(rois), L.ptr(roi_level), lvl, L.ptr(out), B, C, H, W, M, L.f32(sc), res, res,
                    sampling_ratio, L.stream()), "roi_align_rotated_forward_level")
                continue
            L.check(L.lib().jtsm_roi_align_forward_level_f32(
                L.ptr(f), L.ptr(rois), L.ptr(roi_level), lvl, L.ptr(out), B, C, H, W, M, L.f32(sc), res, res,
                sampling_ratio, int(aligned), L.stream()), "roi_align_forward_level")
        ctx.save_for_backward(rois, roi_level)
        ctx.cfg = (res, sampling_ratio, aligned, scales, [tuple(f.shape) for f in feats])
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        rois, roi_level = ctx.saved_tensors
        res, sampling_ratio, aligned, scales, shapes = ctx.cfg
        g = g.contiguous(memory_format=CL)
        nl = len(shapes)
        B, Cc = shapes[0][0], shapes[0][1]
        wanted = [bool(ctx.needs_input_grad[7 + lvl]) for lvl in range(nl)]
        # maps another consumer of the same feature levels already wrote this backward pass (layers/grad_fan.py): add
        # into them — all levels or none, the call has one switch
        sinks = [grad_fan.target(ctx.fans[lvl], shapes[lvl], g.device) if wanted[lvl] else None for lvl in range(nl)]
        accumulate = any(wanted) and all(s is not None for s, w in zip(sinks, wanted) if w)
        grads = [(sinks[lvl] if accumulate else torch.empty(shapes[lvl], dtype=torch.float32, device=g.device,
                                                            memory_format=CL)) if wanted[lvl] else None
                 for lvl in range(nl)]
        if aligned == "rotated":
            # the scatter form, level by level (no planned gather for rotated sample grids)
            for lvl in range(nl):
                if grads[lvl] is None:
                    continue
                L.note_bytes(4.0 * (g.numel() / nl + grads[lvl].numel() + rois.numel()))
                L.check(L.lib().jtsm_roi_align_rotated_backward_level_f32(
                    L.ptr(g), L.ptr(rois), L.ptr(roi_level), lvl, L.ptr(grads[lvl]), B, Cc, shapes[lvl][2],
                    shapes[lvl][3], rois.shape[0], L.f32(scales[lvl]), res, res, sampling_ratio, int(accumulate),
                    L.stream()), "roi_align_rotated_backward_level")
            if accumulate:
                return (None,) * (7 + nl)
            grads = [None if grad_fan.offer(ctx.fans[lvl], grads[lvl]) else grads[lvl] for lvl in range(nl)]
            return (None, None, None, None, None, None, None, *grads)
        Hs = (C.c_int * nl)(*[sh[2] for sh in shapes])
        Ws = (C.c_int * nl)(*[sh[3] for sh in shapes])
        sc = (C.c_float * nl)(*[float(x) for x in scales])
        ptrs = (C.c_void_p * nl)(*[(t.data_ptr() if t is not None else None) for t in grads])
        # read g, write every map
        L.note_bytes(4.0 * (g.numel() + sum(t.numel() for t in grads if t is not None) + rois.numel()))
        lib = L.lib()
        ws = torch.empty(max(lib.jtsm_roi_align_backward_levels_workspace_bytes(Hs, Ws, nl, B, Cc, rois.shape[0]), 16),
                         dtype=torch.uint8, device=g.device)
        L.check(lib.jtsm_roi_align_backward_levels_f32(
            L.ptr(g), L.ptr(rois), L.ptr(roi_level), ptrs, Hs, Ws, sc, nl, B, Cc, rois.shape[0], res, res,
            sampling_ratio, int(aligned), int(accumulate), L.ptr(ws), C.c_size_t(ws.numel()), L.stream()),
            "roi_align_backward_levels")
        if accumulate:
            return (None,) * (7 + nl)
        grads = [None if grad_fan.offer(ctx.fans[lvl], grads[lvl]) else grads[lvl] for lvl in range(nl)]
        return (None, None, None, None, None, None, None, *grads)


class _MOILevels(Function):
    """MOIPool over all FPN levels in one launch each way (csrc/moi_pool.hip: moi_pool_fwd_levels / _bwd_levels)."""

    @staticmethod
    def forward(ctx, rois, roi_level, res, scales, oh_labels, superpixels, fans, *feats):
        ctx.fans = list(fans) if fans is not None else [None] * len(feats)
        feats = [_feat(f) for f in feats]
        M, Cc = rois.shape[0], feats[0].shape[1]
        Lw = oh_labels.shape[1]
        nl = len(feats)
        B = feats[0].shape[0]
        # every roi is on exactly one level, whose kernel writes all of its bins (values and argmax)
        out = torch.empty((M, Cc, res, res), dtype=torch.float32, device=rois.device, memory_format=CL)
        arg = torch.empty((M, Cc, res, res), dtype=torch.int32, device=rois.device, memory_format=CL)
        lib = L.lib()
        Hs = (C.c_int * nl)(*[f.shape[2] for f in feats])
        Ws = (C.c_int * nl)(*[f.shape[3] for f in feats])
        sc = (C.c_float * nl)(*[float(x) for x in scales])
        ptrs = (C.c_void_p * nl)(*[f.data_ptr() for f in feats])
        ws = torch.empty(lib.jtsm_moi_pool_levels_workspace_bytes(B, Hs, Ws, nl, M, Lw), dtype=torch.uint8,
                         device=rois.device)
        # algorithmic bytes: every map, the labels and the superpixels read once; output + argmax written
        L.note_bytes(4.0 * (sum(f.numel() for f in feats) + oh_labels.numel() + superpixels.numel() + 2 * out.numel()))
        L.check(lib.jtsm_moi_pool_forward_levels_f32(
            ptrs, Hs, Ws, sc, nl, L.ptr(rois), L.ptr(roi_level), L.ptr(oh_labels), L.ptr(superpixels), L.ptr(out),
            L.ptr(arg), L.ptr(ws), B, Cc, M, Lw, superpixels.shape[1], superpixels.shape[2], res, res, L.stream()),
            "moi_pool_forward_levels")
        ctx.save_for_backward(rois, roi_level, arg)
        ctx.cfg = (res, [tuple(f.shape) for f in feats], [float(x) for x in scales])
        ctx.mark_non_differentiable(arg)
        # (left on, autograd hands backward a ZERO gradient for the arg-max output: a 50 M-element int32 fill per step)
        ctx.set_materialize_grads(False)
        return out, arg

    @staticmethod
    @once_differentiable
    def backward(ctx, g, _ga=None):
        if g is None:
            return (None,) * (7 + len(ctx.cfg[1]))
        rois, roi_level, arg = ctx.saved_tensors
        res, shapes, scales = ctx.cfg
        g = g.contiguous(memory_format=CL)
        nl = len(shapes)
        B, Cc = shapes[0][0], shapes[0][1]
        wanted = [bool(ctx.needs_input_grad[7 + lvl]) for lvl in range(nl)]
        sinks = [grad_fan.target(ctx.fans[lvl], shapes[lvl], g.device) for lvl in range(nl)]
        accumulate = all(wanted) and all(s is not None for s in sinks)     # (see _AlignLevels.backward)
        grads = sinks if accumulate else [torch.empty(shape, dtype=torch.float32, device=g.device, memory_format=CL)
                                          for shape in shapes]
        Hs = (C.c_int * nl)(*[sh[2] for sh in shapes])
        Ws = (C.c_int * nl)(*[sh[3] for sh in shapes])
        ptrs = (C.c_void_p * nl)(*[t.data_ptr() for t in grads])
        sc = (C.c_float * nl)(*scales)
        lib = L.lib()
        ws = torch.empty(max(lib.jtsm_moi_pool_backward_levels_workspace_bytes(Hs, Ws, nl, B, rois.shape[0]), 16),
                         dtype=torch.uint8, device=g.device)
        L.note_bytes(4.0 * (2 * g.numel() + sum(t.numel() for t in grads)))   # gradient + argmax read, maps written
        L.check(lib.jtsm_moi_pool_backward_levels_f32(
            L.ptr(g), L.ptr(rois), L.ptr(roi_level), L.ptr(arg), ptrs, Hs, Ws, sc, nl, B, Cc, rois.shape[0], res, res,
            int(accumulate), L.ptr(ws), C.c_size_t(ws.numel()), L.stream()), "moi_pool_backward_levels")
        if accumulate:
            return (None,) * (7 + nl)
        grads = [gi if wanted[lvl] else None for lvl, gi in enumerate(grads)]
        grads = [None if grad_fan.offer(ctx.fans[lvl], grads[lvl]) else grads[lvl] for lvl in range(nl)]
        return (None, None, None, None, None, None, None, *grads)


def moi_label_inputs(oh_labels_list, superpixels):
    """The two label operands of MOIPool and of the superpixel-evidence targets in the form the kernels read:
    oh_labels of all images concatenated (rows = the concatenated boxes), padded to a common width, int32; the
    superpixel maps (B,Hs,Ws) int32.  Tensors already in that form pass through."""
    sp = superpixels.tensor if hasattr(superpixels, "tensor") else superpixels
    sp = sp.to(torch.int32).contiguous()
    if isinstance(oh_labels_list, torch.Tensor):
        return oh_labels_list.to(torch.int32).contiguous(), sp
    max_len = max(l.size(1) for l in oh_labels_list)
    labels = cat([l.to(torch.int32) if l.size(1) == max_len else
                  torch.nn.functional.pad(l.to(torch.int32), (0, max_len - l.size(1))) for l in oh_labels_list])
    return labels.contiguous(), sp


class ROIPooler(nn.Module):
    def __init__(self, output_size, scales, sampling_ratio, pooler_type, canonical_box_size=224,
                 canonical_level=4):
        super().__init__()
        if isinstance(output_size, int):
            output_size = (output_size, output_size)
        assert len(output_size) == 2 and isinstance(output_size[0], int) and isinstance(output_size[1], int)
        self.output_size = output_size
        self.pooler_type = pooler_type
        self.sampling_ratio = sampling_ratio
        self.scales = tuple(float(s) for s in scales)
        if pooler_type == "ROIAlign":
            self.level_poolers = nn.ModuleList(
                ROIAlign(output_size, spatial_scale=s, sampling_ratio=sampling_ratio, aligned=False) for s in scales)
        elif pooler_type == "ROIAlignV2":
            self.level_poolers = nn.ModuleList(
                ROIAlign(output_size, spatial_scale=s, sampling_ratio=sampling_ratio, aligned=True) for s in scales)
        elif pooler_type == "MOIPool":
            self.level_poolers = nn.ModuleList(MOIPool(output_size, spatial_scale=s) for s in scales)
        elif pooler_type == "ROIAlignRotated":
            self.level_poolers = nn.ModuleList(
                ROIAlignRotated(output_size, spatial_scale=s, sampling_ratio=sampling_ratio) for s in scales)
        else:
            raise ValueError("Unknown pooler type on the JTSM path: {}".format(pooler_type))
        min_level = -(math.log2(scales[0]))
        max_level = -(math.log2(scales[-1]))
        assert math.isclose(min_level, int(min_level)) and math.isclose(max_level, int(max_level)), \
            "Featuremap stride is not power of 2!"
        self.min_level = int(min_level)
        self.max_level = int(max_level)
        assert len(scales) == self.max_level - self.min_level + 1, \
            "[ROIPooler] Sizes of input featuremaps do not form a pyramid!"
        assert 0 <= self.min_level and self.min_level <= self.max_level
        self.canonical_level = canonical_level
        assert canonical_box_size > 0
        self.canonical_box_size = canonical_box_size

    def forward(self, x: List[torch.Tensor], box_lists, level_ids=None, oh_labels_list=None, superpixels=None):
        """x: per-level (N,C,H,W) maps; box_lists: list[Boxes] per image.  With `superpixels` (an
        ImageList or (N,Hs,Ws) int tensor) and `oh_labels_list`, MOIPool is used and (output, argmax)
        is returned; otherwise the (M,C,P,P) output."""
        num_level_assignments = len(self.level_poolers)
        assert isinstance(x, list) and isinstance(box_lists, list), "Arguments to pooler must be lists"
        assert len(x) == num_level_assignments, \
            "unequal value, num_level_assignments={}, but x is list of {} Tensors".format(num_level_assignments, len(x))
        assert len(box_lists) == x[0].size(0), \
            "unequal value, x[0] batch dim 0 is {}, but box_list has length {}".format(x[0].size(0), len(box_lists))
        if len(box_lists) == 0:
            return torch.zeros((0, x[0].shape[1]) + self.output_size, device=x[0].device, dtype=x[0].dtype)
        # rois and levels in one launch (layers/mining.py: pooler_rois_levels) when the boxes are float32 on the device;
        # the tensor-op helpers above stay as the definition it is tested against and serve every other case
        fused = (level_ids is None and len(box_lists) <= MAX_IMAGES and
                 all(b.tensor.is_cuda and b.tensor.dtype == torch.float32 and b.tensor.shape[-1] == 4 for b in box_lists))
        if fused:
            pooler_fmt_boxes, fused_levels = pooler_rois_levels([b.tensor for b in box_lists], self.min_level, self.max_level,
                                                                self.canonical_box_size, self.canonical_level)
        else:
            pooler_fmt_boxes = convert_boxes_to_pooler_format(box_lists)
        moi = superpixels is not None
        if moi:
            labels, sp = moi_label_inputs(oh_labels_list, superpixels)
        if num_level_assignments == 1:
            if moi:
                return self.level_poolers[0](x[0], pooler_fmt_boxes, labels, sp)
            return self.level_poolers[0](x[0], pooler_fmt_boxes)
        if fused:
            roi_level = fused_levels
        else:
            if level_ids is not None:
                level_assignments = cat(level_ids).to(torch.int64).clamp(0, num_level_assignments - 1)
            else:
                level_assignments = assign_boxes_to_levels(box_lists, self.min_level, self.max_level,
                                                           self.canonical_box_size, self.canonical_level)
            roi_level = level_assignments.to(torch.int32).contiguous()
        rois = pooler_fmt_boxes.to(torch.float32).contiguous()
        fans = [grad_fan.claim(f) for f in x]
        if moi:
            return _MOILevels.apply(rois, roi_level, self.output_size[0], self.scales, labels, sp, fans, *x)
        mode = "rotated" if self.pooler_type == "ROIAlignRotated" else self.pooler_type == "ROIAlignV2"
        return _AlignLevels.apply(rois, roi_level, self.output_size[0], self.sampling_ratio, mode, self.scales, fans, *x)
