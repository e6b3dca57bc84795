#!/usr/bin/env python3
"""The "reference single-GPU images/sec" of BASELINE.md §2: the SAME composite model, inputs, optimizer and
step as bench.py, but with every contraction, normalisation, resampling and loss routed through stock
PyTorch-ROCm operators (F.conv2d / F.linear via MIOpen+hipBLASLt, F.group_norm, F.interpolate,
F.cross_entropy, softmax/BCE), i.e. what the reference's Python does on a GPU.  The three region-pooling
operators keep the HIP kernels: this image has no torchvision and the reference's own CUDA kernels
cannot run here, so no stock implementation exists for them (they are ~2 % of the step).

    python tools/stock_baseline.py [--steps 10 --warmup 3]

Prints one JSON line.  Not part of the product or of bench.py's contract; used for the README table.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CL = torch.channels_last
PAD_ROIS = 0     # --pad-rois: per-roi convolution batches padded with zero rows to a multiple of this (stock run only)


def conv2d_stock(x, w, scale=None, bias=None, residual=None, stride=1, pad=0, dil=1, relu=False,
                 bias_needs_grad=False, emit_planes=True, emit_dx_planes=False):
    n = x.shape[0]
    padded = PAD_ROIS > 0 and n > 16 and n % PAD_ROIS     # (n > 16: a per-roi tensor, not the image batch)
    if padded:
        # MIOpen looks for (and, without a kernel database entry, COMPILES) kernels per exact shape, and the mask
        # heads' batch is the step's foreground count, which changes from step to step on the clustered recipe — the
        # 0.79 images/sec of rounds 2 / 3 was that, not a baseline.  Zero rows up to a multiple of PAD_ROIS keep the
        # set of shapes small; the real rows' results and gradients are unchanged (a convolution is per sample).
        x = F.pad(x, (0, 0, 0, 0, 0, 0, 0, PAD_ROIS - n % PAD_ROIS))
    y = F.conv2d(x, w, None, stride, pad, dil)
    if padded:
        y = y[:n]
    if scale is not None:
        y = y * scale.view(1, -1, 1, 1)
    if bias is not None:
        y = y + bias.view(1, -1, 1, 1)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


def linear_stock(x, w, bias=None, relu=False, bias_needs_grad=True):
    y = F.linear(x, w, bias)
    return F.relu(y) if relu else y


def linear_split_stock(x, weights, biases, relu=False):
    """The predictors as the reference runs them: one F.linear per head."""
    return tuple(linear_stock(x, w, b, relu) for w, b in zip(weights, biases))


def mil_stock(cls_logits, det_logits, bag_offsets, labels, mean_loss=True, max_bag_rows=None):
    off = bag_offsets.tolist()
    counts = [b - a for a, b in zip(off[:-1], off[1:])]
    scores = torch.cat([F.softmax(c, 1) * F.softmax(d, 0)
                        for c, d in zip(cls_logits.split(counts), det_logits.split(counts))])
    probs = torch.cat([s.sum(0, keepdim=True) for s in scores.split(counts)]).clamp(1e-6, 1 - 1e-6)
    loss = F.binary_cross_entropy(probs, labels, reduction="mean" if mean_loss else "sum")
    return (loss if mean_loss else loss / len(counts)), scores.detach(), probs.detach()


def oicr_stock(cls_logits, box_deltas, labels, weights, proposals=None, gt_boxes=None):
    from jtsm_amd.modeling.box_regression import Box2BoxTransform
    w = weights.clone()
    w[labels == -1] = 0
    ce = F.cross_entropy(cls_logits, labels.long(), reduction="none", ignore_index=-1)
    loss_cls = (ce * w).sum() / (w > 1e-12).to(w.dtype).sum()
    if box_deltas is None:
        return loss_cls, loss_cls * 0
    k = cls_logits.shape[1] - 1
    fg = torch.nonzero((labels >= 0) & (labels < k))[:, 0]
    cols = 4 * labels[fg].long()[:, None] + torch.arange(4, device=labels.device)
    tgt = Box2BoxTransform((10.0, 10.0, 5.0, 5.0)).get_deltas(proposals, gt_boxes)
    l1 = (box_deltas[fg[:, None], cols] - tgt[fg]).abs()
    return loss_cls, (l1 * w[fg, None]).sum() / labels.numel()


def patch_to_stock():
    import jtsm_amd.layers.conv as conv
    import jtsm_amd.layers.elementwise as ew
    import jtsm_amd.layers.wrappers as wr
    import jtsm_amd.layers.wsl_losses as wl
    import jtsm_amd.modeling.backbone.fpn as fpn
    import jtsm_amd.modeling.backbone.resnet as resnet
    import jtsm_amd.modeling.roi_heads.fast_rcnn_oicr as oicr
    import jtsm_amd.modeling.roi_heads.fast_rcnn_tsm as tsm
    import jtsm_amd.modeling.roi_heads.roi_heads_jtsm as rh

    import jtsm_amd.layers.fused_blocks as fb

    # every fused autograd node of this repo (bottleneck, mask tower, FC stack) calls the contraction kernels directly:
    # switch them off so each layer goes through conv2d_fused / linear_fused below.  (Before round 2's fix this file
    # left the bottleneck node on, i.e. the "stock" backbone ran this repo's kernels: the figures of round 1 —
    # 11.9-16.5 images/sec — and the first half of round 2 were therefore too HIGH, see profiles/.)
    fb.ENABLED = False
    fb.MASK_TOWER = False
    wr.conv_transpose2x2_ok = lambda x, w: False       # ConvTranspose2d: the GEMM (stock below) + torch pixel shuffle
    conv.conv2d_fused = wr.conv2d_fused = conv2d_stock
    conv.linear_fused = wr.linear_fused = rh.linear_fused = linear_stock
    conv.linear_fused_split = rh.linear_fused_split = linear_split_stock
    resnet.max_pool_3x3_s2 = lambda x: F.max_pool2d(x, 3, 2, 1)
    fpn.upsample2_add = lambda top, lat: lat + F.interpolate(top, scale_factor=2.0, mode="nearest")
    fpn.subsample2 = lambda x: F.max_pool2d(x, 1, 2, 0)
    ew.sum_tensors = lambda xs: sum(xs[1:], xs[0])
    ew.group_norm_relu = lambda x, g, b, groups, eps=1e-5, relu=True: (
        F.relu(F.group_norm(x, groups, g, b, eps)) if relu else F.group_norm(x, groups, g, b, eps))
    ew.upsample_bilinear2x = lambda x: F.interpolate(x, scale_factor=2.0, mode="bilinear", align_corners=False)
    ew.semseg_cross_entropy = lambda z, t, scale=4, ignore_index=255: F.cross_entropy(
        F.interpolate(z, scale_factor=float(scale), mode="bilinear", align_corners=False), t, reduction="mean",
        ignore_index=ignore_index)
    wl.mil_loss = tsm.mil_loss = mil_stock
    wl.oicr_loss = oicr.oicr_loss = oicr_stock


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cluster", type=float, default=1.0, help="as bench.py (0 = the uniform recipe of round 1)")
    ap.add_argument("--objects", type=int, default=40)
    ap.add_argument("--pad-rois", type=int, default=64,
                    help="pad per-roi convolution batches (the mask heads: one row per foreground roi) with zero rows to "
                         "a multiple of this, so that MIOpen meets a handful of shapes instead of a new one per step; "
                         "0 = off (rounds 2 / 3)")
    args = ap.parse_args()
    global PAD_ROIS
    PAD_ROIS = args.pad_rois
    patch_to_stock()
    import bench
    from jtsm_amd.utils.synthetic import synthetic_inputs

    dev = torch.device("cuda", 0)
    model = bench.build(dev)
    inputs = synthetic_inputs(1234, batch=2, size=1024, proposals=2000, device=dev, cluster=args.cluster,
                              objects=args.objects)
    opt = bench.make_optimizer(model, fused=False)

    def step():
        losses = model(inputs)
        total = sum(losses.values())
        total.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        return total

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # self-check: one step with the contraction log on must record NO launch of this repo's contraction kernels
    import jtsm_amd.layers.conv as conv
    conv.LAUNCH_LOG = []
    step()
    torch.cuda.synchronize()
    leaked, conv.LAUNCH_LOG = conv.LAUNCH_LOG, None
    if leaked:
        raise SystemExit("stock_baseline: %d contraction launches of libjtsm_hip.so in the 'stock' step (first: %s)"
                         % (len(leaked), leaked[0][0]))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"what": "stock PyTorch-ROCm operators, same model/step as bench.py (pooling ops: HIP kernels)",
                      "value": round(2 * args.steps / dt, 3), "unit": "images/sec", "ms_per_step": round(1e3 * dt / args.steps, 3),
                      "steps": args.steps, "final_loss": round(float(last.detach()), 5), "cluster": args.cluster,
                      "foreground_rois_last_step": int(model.roi_heads.aux["fg_classes"].numel()),
                      "pad_rois": args.pad_rois, "warmup": args.warmup,
                      "torch": torch.__version__}))


if __name__ == "__main__":
    main()
