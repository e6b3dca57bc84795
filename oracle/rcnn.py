"""ORACLE — TEST INFRASTRUCTURE ONLY.

torch-CPU restatement of the fully supervised Faster R-CNN R50-FPN training step of BASELINE configs[0]
(configs/COCO-Detection/faster_rcnn_R_50_FPN_1x.yaml <- configs/Base-RCNN-FPN.yaml), functional over a flat
{state_dict-name: tensor} dictionary; the backbone and the pooling operators come from oracle/model.py /
oracle/pooling.py.  What each function follows (paths relative to the reference tree):
  cell_anchors / grid_anchors   detectron2/modeling/anchor_generator.py:138-212
  rpn_head                      detectron2/modeling/proposal_generator/rpn.py:64-139
  match                         detectron2/modeling/matcher.py:61-126 (incl. set_low_quality_matches_)
  rpn_labels / rpn_losses       .../rpn.py:283-335,337-412 (smooth_l1 with beta 0 = L1; normaliser batch_size * N)
  rpn_proposals                 .../rpn.py:454-504, proposal_utils.py:12-124
  roi_labels                    detectron2/modeling/roi_heads/roi_heads.py:222-306, proposal_utils.py:127-170
  fast_rcnn_losses              detectron2/modeling/roi_heads/fast_rcnn.py:253-299
Parity unpinned: the reference's tests for these pieces (tests/modeling/test_rpn.py, test_roi_heads.py) pin seeded
RNG-dependent numbers of a torch build that is not this one; what IS pinned by them here: the anchor table of
tests/modeling/test_anchor_generator.py:17-49 and the RPN losses / proposals of test_rpn.py:20-80 are reproduced by
tests/test_oracle_rcnn.py where they do not depend on the RNG.  Random sub-sampling (sampling.py) is the caller's:
with batch sizes >= the candidate counts every labelled element is used and the step is deterministic.
"""
import math

import torch
import torch.nn.functional as F

from . import model as OM
from .inference import batched_nms

NUM_CLASSES = 80
RPN_WEIGHTS, ROI_WEIGHTS = (1.0, 1.0, 1.0, 1.0), (10.0, 10.0, 5.0, 5.0)
SCALE_CLAMP = math.log(1000.0 / 16)
STRIDES = (4, 8, 16, 32, 64)
SIZES = (32, 64, 128, 256, 512)
RATIOS = (0.5, 1.0, 2.0)
PIXEL_MEAN = (103.530, 116.280, 123.675)


def init_params(seed=0, input_gain=1.0, head_gain=1.0):
    """Backbone from oracle/model.py's initialiser; RPN head normal(0.01); box head c2_xavier; predictors
    normal(0.01) / normal(0.001) (rpn.py:100-102, box_head.py:84-87, fast_rcnn.py:403-407)."""
    full = OM.init_params(seed=seed, random_bn=True, input_gain=input_gain)
    p = {k: v for k, v in full.items() if k.startswith("backbone.")}
    g = torch.Generator().manual_seed(seed + 1)
    c, a = 256, len(RATIOS)
    pre = "proposal_generator.rpn_head."
    p[pre + "conv.weight"] = torch.randn(c, c, 3, 3, generator=g) * 0.01
    p[pre + "conv.bias"] = torch.zeros(c)
    p[pre + "objectness_logits.weight"] = torch.randn(a, c, 1, 1, generator=g) * 0.01
    p[pre + "objectness_logits.bias"] = torch.zeros(a)
    p[pre + "anchor_deltas.weight"] = torch.randn(4 * a, c, 1, 1, generator=g) * 0.01
    p[pre + "anchor_deltas.bias"] = torch.zeros(4 * a)
    d_in = c * 7 * 7
    for i in (1, 2):
        bound = math.sqrt(3.0 / d_in)
        # (head_gain: random FPN features are not O(1) the way trained ones are; tests damp fc1 so logits stay O(1))
        p["roi_heads.box_head.fc%d.weight" % i] = (torch.rand(1024, d_in, generator=g) * 2 - 1) * bound * \
            (head_gain if i == 1 else 1.0)
        p["roi_heads.box_head.fc%d.bias" % i] = torch.zeros(1024)
        d_in = 1024
    p["roi_heads.box_predictor.cls_score.weight"] = torch.randn(NUM_CLASSES + 1, d_in, generator=g) * 0.01
    p["roi_heads.box_predictor.cls_score.bias"] = torch.zeros(NUM_CLASSES + 1)
    p["roi_heads.box_predictor.bbox_pred.weight"] = torch.randn(NUM_CLASSES * 4, d_in, generator=g) * 0.001
    p["roi_heads.box_predictor.bbox_pred.bias"] = torch.zeros(NUM_CLASSES * 4)
    p["pixel_mean"] = torch.tensor(PIXEL_MEAN).view(-1, 1, 1)
    p["pixel_std"] = torch.ones(3, 1, 1)
    return p


def trainable_names(p):
    return [k for k in OM.trainable_names(p)] + [k for k in p if k.startswith("proposal_generator.")]


# ----------------------------------------------------------------------------- anchors
def cell_anchors(size, ratios=RATIOS):
    out = []
    area = float(size) ** 2
    for r in ratios:
        w = math.sqrt(area / r)
        h = r * w
        out.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
    return torch.tensor(out, dtype=torch.float32)


def grid_anchors(h, w, stride, cell, offset=0.0):
    xs = torch.arange(offset * stride, w * stride, step=stride, dtype=torch.float32)
    ys = torch.arange(offset * stride, h * stride, step=stride, dtype=torch.float32)
    sy, sx = torch.meshgrid(ys, xs, indexing="ij")
    sh = torch.stack((sx.reshape(-1), sy.reshape(-1), sx.reshape(-1), sy.reshape(-1)), dim=1)
    return (sh.view(-1, 1, 4) + cell.view(1, -1, 4)).reshape(-1, 4)


# ----------------------------------------------------------------------------- box coding
def get_deltas(src, tgt, weights):
    sw, sh = src[:, 2] - src[:, 0], src[:, 3] - src[:, 1]
    sx, sy = src[:, 0] + 0.5 * sw, src[:, 1] + 0.5 * sh
    tw, th = tgt[:, 2] - tgt[:, 0], tgt[:, 3] - tgt[:, 1]
    tx, ty = tgt[:, 0] + 0.5 * tw, tgt[:, 1] + 0.5 * th
    wx, wy, ww, wh = weights
    return torch.stack((wx * (tx - sx) / sw, wy * (ty - sy) / sh, ww * torch.log(tw / sw), wh * torch.log(th / sh)), 1)


def apply_deltas(deltas, boxes, weights):
    w, h = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    cx, cy = boxes[:, 0] + 0.5 * w, boxes[:, 1] + 0.5 * h
    wx, wy, ww, wh = weights
    dx, dy = deltas[:, 0::4] / wx, deltas[:, 1::4] / wy
    dw, dh = (deltas[:, 2::4] / ww).clamp(max=SCALE_CLAMP), (deltas[:, 3::4] / wh).clamp(max=SCALE_CLAMP)
    pcx, pcy = dx * w[:, None] + cx[:, None], dy * h[:, None] + cy[:, None]
    pw, ph = torch.exp(dw) * w[:, None], torch.exp(dh) * h[:, None]
    out = torch.zeros_like(deltas)
    out[:, 0::4], out[:, 1::4], out[:, 2::4], out[:, 3::4] = pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph
    return out


def match(iou, thresholds, labels, low_quality):
    """iou (G, N) -> (matched gt index (N,), label (N,))."""
    n = iou.shape[1]
    if iou.numel() == 0:
        return torch.zeros(n, dtype=torch.int64), torch.full((n,), labels[0], dtype=torch.int8)
    val, idx = iou.max(dim=0)
    lab = torch.full((n,), 1, dtype=torch.int8)
    cuts = [-float("inf")] + list(thresholds) + [float("inf")]
    for l, lo, hi in zip(labels, cuts[:-1], cuts[1:]):
        lab[(val >= lo) & (val < hi)] = l
    if low_quality:
        best = iou.max(dim=1, keepdim=True).values
        lab[(iou == best).any(dim=0)] = 1
    return idx, lab


# ----------------------------------------------------------------------------- the step
def forward_losses(p, batch, rpn_batch=256, roi_batch=512, pre_nms_topk=2000, post_nms_topk=1000, return_aux=False):
    """batch: dict(images=[(3,H,W)], gt_boxes=[(G_i,4)], gt_classes=[(G_i,) int64]).  rpn_batch / roi_batch must be
    at least the number of labelled anchors / proposals per image (no random sub-sampling is restated)."""
    x = OM.preprocess(p, batch["images"], 32)
    sizes = [tuple(im.shape[-2:]) for im in batch["images"]]
    N = x.shape[0]
    feats = OM.resnet_fpn(p, x, 50)
    levels = [feats["p%d" % l] for l in (2, 3, 4, 5, 6)]

    # ---- RPN head and anchors
    pre = "proposal_generator.rpn_head."
    anchors, logits, deltas = [], [], []
    for f, stride, size in zip(levels, STRIDES, SIZES):
        t = F.relu(F.conv2d(f, p[pre + "conv.weight"], p[pre + "conv.bias"], 1, 1))
        z = F.conv2d(t, p[pre + "objectness_logits.weight"], p[pre + "objectness_logits.bias"])
        d = F.conv2d(t, p[pre + "anchor_deltas.weight"], p[pre + "anchor_deltas.bias"])
        logits.append(z.permute(0, 2, 3, 1).flatten(1))
        deltas.append(d.view(N, -1, 4, d.shape[-2], d.shape[-1]).permute(0, 3, 4, 1, 2).flatten(1, -2))
        anchors.append(grid_anchors(f.shape[2], f.shape[3], stride, cell_anchors(size)))
    A = torch.cat(anchors)
    all_logits, all_deltas = torch.cat(logits, 1), torch.cat(deltas, 1)

    # ---- RPN labels and losses
    labels, matched = [], []
    for gt in batch["gt_boxes"]:
        idx, lab = match(OM.pairwise_iou(gt, A), (0.3, 0.7), (0, -1, 1), True)
        assert int((lab == 1).sum()) <= rpn_batch // 2 and int((lab == 0).sum()) <= rpn_batch - int((lab == 1).sum()), \
            "rpn_batch too small: sub-sampling would be random"
        labels.append(lab.to(torch.int64))
        matched.append(gt[idx] if len(gt) else torch.zeros_like(A))
    L = torch.stack(labels)
    pos, valid = L == 1, L >= 0
    tgt = torch.stack([get_deltas(A, m, RPN_WEIGHTS) for m in matched])
    losses = {"loss_rpn_cls": F.binary_cross_entropy_with_logits(all_logits[valid], L[valid].to(torch.float32),
                                                                 reduction="sum") / (rpn_batch * N),
              "loss_rpn_loc": (all_deltas[pos] - tgt[pos]).abs().sum() / (rpn_batch * N)}

    # ---- proposals
    proposals = []
    with torch.no_grad():
        bl, sl, ll = [], [], []
        for lvl, (a, z, d) in enumerate(zip(anchors, logits, deltas)):
            dec = apply_deltas(d.reshape(-1, 4), a.unsqueeze(0).expand(N, -1, -1).reshape(-1, 4), RPN_WEIGHTS).view(N, -1, 4)
            k = min(pre_nms_topk, z.shape[1])
            top, idx = z.sort(descending=True, dim=1)
            bl.append(torch.gather(dec, 1, idx[:, :k].unsqueeze(2).expand(-1, -1, 4)))
            sl.append(top[:, :k])
            ll.append(torch.full((k,), lvl, dtype=torch.int64))
        B, S, LV = torch.cat(bl, 1), torch.cat(sl, 1), torch.cat(ll)
        for n, (h, w) in enumerate(sizes):
            b = B[n].clone()
            b[:, 0::2] = b[:, 0::2].clamp(0, w)
            b[:, 1::2] = b[:, 1::2].clamp(0, h)
            ok = ((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0)
            b, s, lv = b[ok], S[n][ok], LV[ok]
            keep = batched_nms(b, s, lv, 0.7)[:post_nms_topk]
            proposals.append((b[keep], s[keep]))

    # ---- ROI heads: append gt, match, (no sub-sampling), pool, box head, losses
    rois, cls_l, gtb_l = [], [], []
    for n, ((b, _), gt, gc) in enumerate(zip(proposals, batch["gt_boxes"], batch["gt_classes"])):
        b = torch.cat([b, gt])
        idx, lab = match(OM.pairwise_iou(gt, b), (0.5,), (0, 1), False)
        if len(gt):
            cls = gc[idx].clone()
            cls[lab == 0] = NUM_CLASSES
        else:
            cls = torch.full((len(b),), NUM_CLASSES, dtype=torch.int64)
        nfg = int((cls != NUM_CLASSES).sum())
        assert nfg <= roi_batch // 4 and len(b) - nfg <= roi_batch - nfg, "roi_batch too small: sub-sampling would be random"
        rois.append(torch.cat([torch.full((len(b), 1), float(n)), b], 1))
        cls_l.append(cls)
        gtb_l.append(gt[idx] if len(gt) else b)
    rois, cls, gtb = torch.cat(rois), torch.cat(cls_l), torch.cat(gtb_l)
    pooled = OM.roi_align_levels(levels[:4], rois, x.shape[2], 7)
    h = pooled.flatten(1)
    for i in (1, 2):
        h = F.relu(F.linear(h, p["roi_heads.box_head.fc%d.weight" % i], p["roi_heads.box_head.fc%d.bias" % i]))
    scores = F.linear(h, p["roi_heads.box_predictor.cls_score.weight"], p["roi_heads.box_predictor.cls_score.bias"])
    dl = F.linear(h, p["roi_heads.box_predictor.bbox_pred.weight"], p["roi_heads.box_predictor.bbox_pred.bias"])
    losses["loss_cls"] = F.cross_entropy(scores, cls, reduction="mean")
    fg = torch.nonzero(cls < NUM_CLASSES)[:, 0]
    picked = dl.view(-1, NUM_CLASSES, 4)[fg, cls[fg]]
    losses["loss_box_reg"] = (picked - get_deltas(rois[fg, 1:], gtb[fg], ROI_WEIGHTS)).abs().sum() / max(cls.numel(), 1)
    if return_aux:
        return losses, dict(proposals=proposals, rpn_labels=L, roi_classes=cls, rois=rois)
    return losses


def synthetic_batch(seed, B=2, h=128, w=160, n_gt=4):
    """utils/testing.py:29-40 recipe (random boxes, at least 16 px a side here), classes U{0..79}."""
    g = torch.Generator().manual_seed(seed)
    out = dict(images=[], gt_boxes=[], gt_classes=[])
    for _ in range(B):
        out["images"].append(torch.rand(3, h, w, generator=g) * 255)
        x0 = torch.rand(n_gt, generator=g) * (w - 40)
        y0 = torch.rand(n_gt, generator=g) * (h - 40)
        bw = torch.rand(n_gt, generator=g) * (w - 40) * 0.6 + 16
        bh = torch.rand(n_gt, generator=g) * (h - 40) * 0.6 + 16
        out["gt_boxes"].append(torch.stack([x0, y0, (x0 + bw).clamp(max=w), (y0 + bh).clamp(max=h)], 1))
        out["gt_classes"].append(torch.randint(0, NUM_CLASSES, (n_gt,), generator=g))
    return out
