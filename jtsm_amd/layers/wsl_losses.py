"""Fused MIL / OICR losses (jtsm_amd/csrc/wsl_losses.hip) as autograd functions.

mil_loss      <- TSMOutputLayers.forward + TSMOutputs.binary_cross_entropy_loss
                 (projects/WSL/wsl/modeling/roi_heads/fast_rcnn_tsm.py:548-598,346-379)
oicr_loss     <- OICROutputs.softmax_cross_entropy_loss + box_reg_loss("smooth_l1_weighted")
                 (projects/WSL/wsl/modeling/roi_heads/fast_rcnn_oicr.py:282-298,350-380)
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib as L


def _rowmajor(t):
    """(tensor, leading dimension) of a 2-D float32 view whose rows are unit-stride."""
    if t.dim() != 2 or t.dtype != torch.float32:
        raise RuntimeError("expected a 2-D float32 tensor, got %s %s" % (tuple(t.shape), t.dtype))
    if t.stride(1) != 1 or t.stride(0) < t.shape[1]:
        t = t.contiguous()
    return t, t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


class _MILLoss(Function):
    @staticmethod
    def forward(ctx, cls_logits, det_logits, bag_offsets, labels, mean_loss, max_bag_rows):
        L.require_gpu(cls_logits, det_logits, bag_offsets, labels)
        c, ldc = _rowmajor(cls_logits)
        d, ldd = _rowmajor(det_logits)
        if ldc != ldd:
            c, d = c.contiguous(), d.contiguous()
            ldc = ldd = c.shape[1]
        R, nc = c.shape
        nb = bag_offsets.numel() - 1
        labels = labels.to(torch.float32).contiguous()
        scores = torch.empty((R, nc), dtype=torch.float32, device=c.device)
        probs = torch.empty((nb, nc), dtype=torch.float32, device=c.device)
        loss = torch.empty((), dtype=torch.float32, device=c.device)
        ws = torch.empty(L.lib().jtsm_mil_workspace_bytes(nb, max_bag_rows, nc), dtype=torch.uint8, device=c.device)
        L.check(L.lib().jtsm_mil_forward_f32(L.ptr(c), L.ptr(d), ldc, nc, L.ptr(bag_offsets), nb, max_bag_rows,
                                             L.ptr(labels), int(bool(mean_loss)), L.ptr(scores), L.ptr(probs),
                                             L.ptr(loss), L.ptr(ws), L.stream()), "mil_forward")
        ctx.save_for_backward(c, d, bag_offsets, ws)
        ctx.cfg = (ldc, nc, nb, max_bag_rows)
        ctx.mark_non_differentiable(scores, probs)
        ctx.set_materialize_grads(False)   # no zero-filled "gradients" for the two non-differentiable outputs
        return loss, scores, probs

    @staticmethod
    @once_differentiable
    def backward(ctx, g_loss, _gs, _gp):
        if g_loss is None:
            return None, None, None, None, None, None
        c, d, bag_offsets, ws = ctx.saved_tensors
        ldc, nc, nb, mbr = ctx.cfg
        dc = torch.empty((c.shape[0], nc), dtype=torch.float32, device=c.device)
        dd = torch.empty_like(dc)
        g = g_loss.to(torch.float32).contiguous()
        L.check(L.lib().jtsm_mil_backward_f32(L.ptr(c), L.ptr(d), ldc, nc, L.ptr(bag_offsets), nb, mbr, L.ptr(g),
                                              L.ptr(dc), L.ptr(dd), nc, L.ptr(ws), L.stream()), "mil_backward")
        return dc, dd, None, None, None, None


def mil_loss(cls_logits, det_logits, bag_offsets, labels, mean_loss=True, max_bag_rows=None):
    """Returns (loss, scores (R,nc), img_probs (B,nc)); only `loss` carries gradient.
    bag_offsets: int32 (B+1,) device tensor; max_bag_rows: host int upper bound of rows per image."""
    if max_bag_rows is None:
        max_bag_rows = int(cls_logits.shape[0])
    return _MILLoss.apply(cls_logits, det_logits, bag_offsets, labels, mean_loss, int(max_bag_rows))


class _MILScores(Function):
    """scores = softmax_c(C) * per-bag softmax_r(D) with gradient to both logits (the reference-shaped
    TSMOutputLayers.forward).  Forward: the HIP kernel; backward of an arbitrary upstream G:
        dC = p (G q - sum_c G q p),   dD = q (G p - sum_{r in bag} G p q)      (p, q the two softmaxes)."""

    @staticmethod
    def forward(ctx, cls_logits, det_logits, bag_offsets, max_bag_rows):
        L.require_gpu(cls_logits, det_logits, bag_offsets)
        c, ldc = _rowmajor(cls_logits)
        d, ldd = _rowmajor(det_logits)
        if ldc != ldd:
            c, d = c.contiguous(), d.contiguous()
            ldc = c.shape[1]
        R, nc = c.shape
        nb = bag_offsets.numel() - 1
        scores = torch.empty((R, nc), dtype=torch.float32, device=c.device)
        probs = torch.empty((nb, nc), dtype=torch.float32, device=c.device)
        loss = torch.empty((), dtype=torch.float32, device=c.device)
        labels = torch.zeros((nb, nc), dtype=torch.float32, device=c.device)
        ws = torch.empty(L.lib().jtsm_mil_workspace_bytes(nb, max_bag_rows, nc), dtype=torch.uint8, device=c.device)
        L.check(L.lib().jtsm_mil_forward_f32(L.ptr(c), L.ptr(d), ldc, nc, L.ptr(bag_offsets), nb, max_bag_rows,
                                             L.ptr(labels), 1, L.ptr(scores), L.ptr(probs), L.ptr(loss), L.ptr(ws),
                                             L.stream()), "mil_forward")
        ctx.save_for_backward(c, scores, bag_offsets)
        return scores

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        c, scores, bag_offsets = ctx.saved_tensors
        p = torch.softmax(c, dim=1)
        q = scores / p.clamp_min(1e-38)
        gs = g * scores                                              # G p q
        dc = g * scores - p * gs.sum(dim=1, keepdim=True)            # p (G q - sum_c G q p)
        bag_of = torch.bucketize(torch.arange(c.shape[0], device=c.device), bag_offsets[1:].to(torch.int64), right=True)
        per_bag = torch.zeros((bag_offsets.numel() - 1, c.shape[1]), dtype=gs.dtype, device=gs.device).index_add_(0, bag_of, gs)
        dd = gs - q * per_bag[bag_of]                                # q (G p - sum_r G p q)
        return dc, dd, None, None


def mil_scores(cls_logits, det_logits, counts):
    """MIL scores (R, nc) of ragged bags (`counts` rows per image), differentiable."""
    offs = [0]
    for n in counts:
        offs.append(offs[-1] + int(n))
    bag_offsets = torch.tensor(offs, dtype=torch.int32).to(cls_logits.device, non_blocking=True)
    return _MILScores.apply(cls_logits, det_logits, bag_offsets, max(max(counts), 1))


class _OICRLoss(Function):
    @staticmethod
    def forward(ctx, cls_logits, box_deltas, labels, weights, proposals, gt_boxes):
        L.require_gpu(cls_logits, labels, weights)
        z, ldz = _rowmajor(cls_logits)
        R, ncls = z.shape
        has_box = box_deltas is not None
        if has_box:
            dl, ldd = _rowmajor(box_deltas)
            proposals, gt_boxes = proposals.to(torch.float32).contiguous(), gt_boxes.to(torch.float32).contiguous()
        else:
            dl, ldd = None, 0
        labels = labels.to(torch.int32).contiguous()
        weights = weights.to(torch.float32).contiguous()
        out = torch.empty(4, dtype=torch.float32, device=z.device)
        ws = torch.empty(L.lib().jtsm_oicr_workspace_bytes(), dtype=torch.uint8, device=z.device)
        L.check(L.lib().jtsm_oicr_forward_f32(L.ptr(z), ldz, ncls, L.ptr(dl), ldd, L.ptr(labels), L.ptr(weights),
                                              L.ptr(proposals if has_box else None),
                                              L.ptr(gt_boxes if has_box else None), R, L.ptr(out), L.ptr(ws),
                                              L.stream()), "oicr_forward")
        ctx.save_for_backward(z, dl, labels, weights, proposals if has_box else None,
                              gt_boxes if has_box else None, out)
        ctx.cfg = (ldz, ncls, ldd, R, has_box)
        return out[0], out[1]

    @staticmethod
    @once_differentiable
    def backward(ctx, g_cls, g_box):
        z, dl, labels, weights, proposals, gt_boxes, out = ctx.saved_tensors
        ldz, ncls, ldd, R, has_box = ctx.cfg
        dz = torch.empty((R, ncls), dtype=torch.float32, device=z.device)
        dd = torch.empty((R, 4 * (ncls - 1)), dtype=torch.float32, device=z.device) if has_box else None
        gc = g_cls.to(torch.float32).contiguous()
        gb = g_box.to(torch.float32).contiguous() if has_box else None
        L.check(L.lib().jtsm_oicr_backward_f32(
            L.ptr(z), ldz, ncls, L.ptr(dl), ldd, L.ptr(labels), L.ptr(weights), L.ptr(proposals), L.ptr(gt_boxes),
            R, L.ptr(out), L.ptr(gc), L.ptr(gb), L.ptr(dz), ncls, L.ptr(dd), 4 * (ncls - 1), L.stream()),
            "oicr_backward")
        return dz, dd, None, None, None, None


def oicr_loss(cls_logits, box_deltas, labels, weights, proposals=None, gt_boxes=None):
    """(loss_cls, loss_box_reg) of one refinement branch.  box_deltas may be None."""
    return _OICRLoss.apply(cls_logits, box_deltas, labels, weights, proposals, gt_boxes)


class _MaskBCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, gt_classes, target):
        L.require_gpu(logits, target)
        n, c, side, side2 = logits.shape
        if side != side2 or tuple(target.shape) != (n, side, side):
            raise RuntimeError("mask_bce: logits %s / target %s" % (tuple(logits.shape), tuple(target.shape)))
        # channels_last storage with any channel pitch (a [:, :80] slice of a padded map is fine)
        if not (logits.stride(1) == 1 and logits.stride(3) >= c and logits.stride(2) == side * logits.stride(3)
                and logits.stride(0) == side * logits.stride(2)):
            logits = logits.contiguous(memory_format=torch.channels_last)
        ld = logits.stride(3)
        target = target.to(torch.uint8).contiguous()
        cls = gt_classes.to(torch.int64).contiguous() if c > 1 else None
        lib = L.lib()
        out = torch.empty(1, dtype=torch.float32, device=logits.device)
        ws = torch.empty(lib.jtsm_mask_bce_workspace_bytes(), dtype=torch.uint8, device=logits.device)
        L.check(lib.jtsm_mask_bce_forward_f32(L.ptr(logits), ld, c, L.ptr(cls), L.ptr(target), n, side, L.ptr(out),
                                              L.ptr(ws), L.stream()), "mask_bce_forward")
        ctx.save_for_backward(logits, cls, target)
        ctx.cfg = (ld, c, n, side)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        logits, cls, target = ctx.saved_tensors
        ld, c, n, side = ctx.cfg
        d = torch.empty((n, side, side, ld), dtype=torch.float32, device=logits.device)
        g = g.to(torch.float32).contiguous()
        L.check(L.lib().jtsm_mask_bce_backward_f32(L.ptr(logits), ld, c, L.ptr(cls), L.ptr(target), n, side, L.ptr(g),
                                                   L.ptr(d), L.stream()), "mask_bce_backward")
        return d.permute(0, 3, 1, 2)[:, :c], None, None


def mask_bce_loss(logits, gt_classes, target):
    """F.binary_cross_entropy_with_logits(logits[arange(N), gt_classes], target.float(), reduction="mean") in one
    launch each way (libjtsm_hip.so: mask_bce_*)."""
    return _MaskBCE.apply(logits, gt_classes, target)
