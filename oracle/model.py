"""ORACLE — TEST INFRASTRUCTURE ONLY.

torch-CPU restatement of the R50-FPN JTSM composite's training step (the composite is
Base-RCNN-FPN backbone + GeneralizedMCNNWSL + JTSMROIHeads + SemSegFPNHead), written functionally
over a flat {state_dict-name: tensor} dictionary (names: SURVEY Appendix C) with the stock torch
operators the reference composes, plus oracle/pooling.py for the three pooling operators.  It is the
checker for jtsm_amd's HIP path and the `cpu_baseline` ("port") of bench.py.  Plain NCHW tensors.

What each function follows (paths relative to the reference tree):
  preprocess            projects/WSL/wsl/modeling/meta_arch/mcnn.py:303-318; structures/image_list.py:71-125
  frozen_bn / conv      detectron2/layers/batch_norm.py:45-66; layers/wrappers.py:62-83
  resnet_fpn            modeling/backbone/resnet.py:101-211,331-359,424-447; fpn.py:114-152,173-185
  assign_levels         modeling/poolers.py:22-58
  moi_pool_levels       projects/WSL/wsl/modeling/poolers.py:261-320 (multi-level semantics DEFINED here, F2)
  box_branch            projects/WSL/wsl/modeling/roi_heads/roi_heads_jtsm.py:590-737
  mil_scores/mil_loss   .../fast_rcnn_tsm.py:548-598,346-379,840-854
  oicr_losses           .../fast_rcnn_oicr.py:243-247,282-298,300-380
  mine_top1             .../roi_heads_jtsm.py:1167-1338 (top_k=1, thres=0 path)
  match_and_label       .../roi_heads.py:222-370; detectron2/modeling/matcher.py:61-103; structures/boxes.py:345-392
  box_deltas/apply      detectron2/modeling/box_regression.py:38-113
  mask_branch           .../roi_heads_jtsm.py:754-948; .../mask_head.py:23-103,266-343; structures/masks.py:169-200
  semseg_head           detectron2/modeling/meta_arch/semantic_seg.py:103-188
  pgt_sem_seg           .../roi_heads_jtsm.py:2025-2070 (masks: need_mask=True -> :1333-1334 -> :1928-1994)
  near targets          .../roi_heads_jtsm.py:840-905 (top-10 foreground proposals per pseudo box, second labelling)
  evidence masks        .../roi_heads_jtsm.py:1928-1994 (object_evidence, superpixel branch)
  refinery targets      .../roi_heads_jtsm.py:1997-2022 (get_pgt_mask: paste, then crop_and_resize)
Declared substitutions (also made by the product path): grabCut (object_evidence's live branch, OpenCV) is
replaced by the reference's own superpixel-evidence construction; masks stay bitmasks where the reference
encodes them as polygons and rasterises those again.  The pseudo SEMANTIC target is painted from the targets'
superpixel-evidence masks (:2038-2069).  mask_targets="rect" / sem_targets="rect" select round 1's rectangle masks
(+ thresholded refinery targets / rectangles eroded by 2 px in the semantic target).
Dropout is a caller-supplied mask (None = off).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import pooling as P

NUM_THINGS = 80
NUM_STUFF = 54           # MODEL.SEM_SEG_HEAD.NUM_CLASSES (0 = "things", 1..53 stuff)
NUM_MIL = NUM_THINGS + NUM_STUFF - 1
STAGES = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}
PIXEL_MEAN = (102.9801, 115.9465, 122.7717)


# ----------------------------------------------------------------------------- parameters
def init_params(seed=0, depth=50, dan_dims=(2048, 4096), refine_k=4, random_bn=True, input_gain=1.0):
    """Seeded synthetic weights with the reference's initialisers.
    random_bn=True perturbs the FrozenBN buffers away from identity so scale/bias paths are exercised.
    input_gain scales the stem convolution: random weights are not matched to 0..255 pixel inputs the way
    pretrained ones are, and 1/64 keeps activations O(1) so softmaxes do not saturate in parity tests."""
    g = torch.Generator().manual_seed(seed)
    p = {}

    def msra(name, o, i, k):
        p[name] = torch.randn(o, i, k, k, generator=g) * math.sqrt(2.0 / (o * k * k))

    def bn(name, c):
        if random_bn:
            p[name + ".weight"] = torch.rand(c, generator=g) * 0.5 + 0.75
            p[name + ".bias"] = torch.randn(c, generator=g) * 0.1
            p[name + ".running_mean"] = torch.randn(c, generator=g) * 0.1
            p[name + ".running_var"] = torch.rand(c, generator=g) * 0.5 + 0.75
        else:
            p[name + ".weight"], p[name + ".bias"] = torch.ones(c), torch.zeros(c)
            p[name + ".running_mean"], p[name + ".running_var"] = torch.zeros(c), torch.ones(c)

    def xavier(name, o, i, k):  # c2_xavier_fill: kaiming_uniform(a=1), zero bias
        bound = math.sqrt(3.0 / (i * k * k))
        p[name + ".weight"] = (torch.rand(o, i, k, k, generator=g) * 2 - 1) * bound
        p[name + ".bias"] = torch.zeros(o)

    bu = "backbone.bottom_up."
    msra(bu + "stem.conv1.weight", 64, 3, 7)
    p[bu + "stem.conv1.weight"] *= input_gain
    bn(bu + "stem.conv1.norm", 64)
    cin = 64
    for si, nblocks in enumerate(STAGES[depth]):
        stage, mid, cout = "res%d" % (si + 2), 64 * 2 ** si, 256 * 2 ** si
        for b in range(nblocks):
            pre = "%s%s.%d." % (bu, stage, b)
            if cin != cout:
                msra(pre + "shortcut.weight", cout, cin, 1)
                bn(pre + "shortcut.norm", cout)
            msra(pre + "conv1.weight", mid, cin, 1)
            bn(pre + "conv1.norm", mid)
            msra(pre + "conv2.weight", mid, mid, 3)
            bn(pre + "conv2.norm", mid)
            msra(pre + "conv3.weight", cout, mid, 1)
            bn(pre + "conv3.norm", cout)
            cin = cout
    for lvl, c in zip((2, 3, 4, 5), (256, 512, 1024, 2048)):
        xavier("backbone.fpn_lateral%d" % lvl, 256, c, 1)
        xavier("backbone.fpn_output%d" % lvl, 256, 256, 3)

    rh = "roi_heads."
    d_in = 256 * 7 * 7
    for i, d in enumerate(dan_dims):
        p["%sbox_head.fc%d.weight" % (rh, i + 1)] = torch.randn(d, d_in, generator=g) * 0.005
        p["%sbox_head.fc%d.bias" % (rh, i + 1)] = torch.full((d,), 0.1)
        d_in = d
    for nm in ("cls", "det"):
        bound = math.sqrt(6.0 / (d_in + NUM_MIL))
        p["%sbox_predictor.%s.weight" % (rh, nm)] = (torch.rand(NUM_MIL, d_in, generator=g) * 2 - 1) * bound
        p["%sbox_predictor.%s.bias" % (rh, nm)] = torch.zeros(NUM_MIL)
    for k in range(refine_k):
        p["%sbox_refinery_%d.cls_score.weight" % (rh, k)] = torch.randn(NUM_THINGS + 1, d_in, generator=g) * 0.01
        p["%sbox_refinery_%d.cls_score.bias" % (rh, k)] = torch.zeros(NUM_THINGS + 1)
        p["%sbox_refinery_%d.bbox_pred.weight" % (rh, k)] = torch.randn(NUM_THINGS * 4, d_in, generator=g) * 0.001
        p["%sbox_refinery_%d.bbox_pred.bias" % (rh, k)] = torch.zeros(NUM_THINGS * 4)
    for head in ("mask_head", "mask_refinery_0"):
        for k in range(4):
            msra("%s%s.mask_fcn%d.weight" % (rh, head, k + 1), 256, 256, 3)
            p["%s%s.mask_fcn%d.bias" % (rh, head, k + 1)] = torch.zeros(256)
        # ConvTranspose2d weight is (in, out, 2, 2); c2_msra_fill -> fan_out = in*k*k for this layout
        p["%s%s.deconv.weight" % (rh, head)] = torch.randn(256, 256, 2, 2, generator=g) * math.sqrt(2.0 / (256 * 4))
        p["%s%s.deconv.bias" % (rh, head)] = torch.zeros(256)
        p["%s%s.predictor.weight" % (rh, head)] = torch.randn(NUM_THINGS, 256, 1, 1, generator=g) * 0.001
        p["%s%s.predictor.bias" % (rh, head)] = torch.zeros(NUM_THINGS)
    for lvl, nconv in zip((2, 3, 4, 5), (1, 1, 2, 3)):
        for j in range(nconv):
            # nn.Sequential index: conv modules sit at even slots when an Upsample follows each conv
            idx = j * (1 if lvl == 2 else 2)
            msra("sem_seg_head.p%d.%d.weight" % (lvl, idx), 128, 256 if j == 0 else 128, 3)
            p["sem_seg_head.p%d.%d.norm.weight" % (lvl, idx)] = torch.ones(128)
            p["sem_seg_head.p%d.%d.norm.bias" % (lvl, idx)] = torch.zeros(128)
    msra("sem_seg_head.predictor.weight", NUM_STUFF, 128, 1)
    p["sem_seg_head.predictor.bias"] = torch.zeros(NUM_STUFF)
    p["pixel_mean"] = torch.tensor(PIXEL_MEAN).view(-1, 1, 1)
    p["pixel_std"] = torch.ones(3, 1, 1)
    return p


TRAINABLE_PREFIXES = ("backbone.bottom_up.res3", "backbone.bottom_up.res4", "backbone.bottom_up.res5",
                      "backbone.fpn_", "roi_heads.", "sem_seg_head.")


def trainable_names(p):
    """FREEZE_AT=2 (defaults.py:130): stem+res2 frozen; FrozenBN buffers and pixel stats never train."""
    out = []
    for k in p:
        if ".norm.running_" in k or k.startswith("pixel_"):
            continue
        if ".norm." in k and k.startswith("backbone.bottom_up"):
            continue  # FrozenBatchNorm2d affine terms are buffers
        if k.startswith(TRAINABLE_PREFIXES):
            out.append(k)
    return out


# ----------------------------------------------------------------------------- backbone
def preprocess(p, images, divisibility=32):
    """images: list of (3,H,W) float tensors -> (N,3,Hp,Wp) normalised, zero padded bottom/right."""
    xs = [(im - p["pixel_mean"]) / p["pixel_std"] for im in images]
    H = max(x.shape[1] for x in xs)
    W = max(x.shape[2] for x in xs)
    H = (H + divisibility - 1) // divisibility * divisibility
    W = (W + divisibility - 1) // divisibility * divisibility
    out = xs[0].new_zeros(len(xs), 3, H, W)
    for i, x in enumerate(xs):
        out[i, :, : x.shape[1], : x.shape[2]] = x
    return out


def frozen_bn(p, name, x, eps=1e-5):
    scale = p[name + ".weight"] * (p[name + ".running_var"] + eps).rsqrt()
    bias = p[name + ".bias"] - p[name + ".running_mean"] * scale
    return x * scale.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)


# Forced discrete choices (tests/test_hip_model.py: whole-step gradients at 1e-4): when set, ReLU gates named here,
# the MOIPool winners and the refinery's mask targets are TAKEN from this dict instead of being decided by this
# run's own (rounding-dependent) values, so that two implementations differentiate the same piecewise-linear map.
_FORCED = None


def gated_relu(y, name):
    if _FORCED is not None and name in _FORCED.get("gates", {}):
        return y * _FORCED["gates"][name].to(y.dtype)
    return F.relu(y)


def conv_bn(p, name, x, stride=1, pad=0, relu=False, dil=1):
    y = frozen_bn(p, name + ".norm", F.conv2d(x, p[name + ".weight"], None, stride, pad, dil))
    return gated_relu(y, name) if relu else y


def bottleneck(p, pre, x, stride):
    """STRIDE_IN_1X1=True (defaults.py:479): the stride sits in conv1 and the shortcut."""
    out = conv_bn(p, pre + "conv1", x, stride, 0, True)
    out = conv_bn(p, pre + "conv2", out, 1, 1, True)
    out = conv_bn(p, pre + "conv3", out)
    sc = conv_bn(p, pre + "shortcut", x, stride) if (pre + "shortcut.weight") in p else x
    return gated_relu(out + sc, pre + "conv3")


def resnet_fpn(p, x, depth=50):
    bu = "backbone.bottom_up."
    x = conv_bn(p, bu + "stem.conv1", x, 2, 3, True)
    x = F.max_pool2d(x, 3, 2, 1)
    feats = {}
    for si, nblocks in enumerate(STAGES[depth]):
        stage = "res%d" % (si + 2)
        for b in range(nblocks):
            x = bottleneck(p, "%s%s.%d." % (bu, stage, b), x, 2 if (b == 0 and si > 0) else 1)
        feats[stage] = x
    out, prev = {}, None
    for lvl in (5, 4, 3, 2):
        lat = F.conv2d(feats["res%d" % lvl], p["backbone.fpn_lateral%d.weight" % lvl],
                       p["backbone.fpn_lateral%d.bias" % lvl])
        prev = lat if prev is None else lat + F.interpolate(prev, scale_factor=2.0, mode="nearest")
        out["p%d" % lvl] = F.conv2d(prev, p["backbone.fpn_output%d.weight" % lvl],
                                    p["backbone.fpn_output%d.bias" % lvl], 1, 1)
    out["p6"] = F.max_pool2d(out["p5"], 1, 2, 0)
    return out


# ----------------------------------------------------------------------------- ResNet-WS v2, dilated C5 (shipped configs)
STAGES_WSR = {18: (2, 2, 2, 2), 34: (3, 4, 6, 3), 50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}


def init_params_dc5(seed=0, depth=18, nt=20, ns=2, dan_dims=(4096, 4096), refine_k=4, random_bn=True, input_gain=1.0):
    """Seeded weights of the shipped single-level composite (projects/WSL/configs/PascalVOC-PanopticSegmentation/
    jtsm_WSR_{18,50}_DC5_1x.yaml): ResNet-WS v2 with dilated res4/res5 (resnet_wsl_v2.py), DAN, MIL + refinements,
    two mask heads on res5; the TwoClassHead has no parameters.  state_dict names as the product's."""
    g = torch.Generator().manual_seed(seed)
    p = {}

    def msra(name, o, i, k):
        p[name] = torch.randn(o, i, k, k, generator=g) * math.sqrt(2.0 / (o * k * k))

    def bn(name, c):
        if random_bn:
            p[name + ".weight"] = torch.rand(c, generator=g) * 0.5 + 0.75
            p[name + ".bias"] = torch.randn(c, generator=g) * 0.1
            p[name + ".running_mean"] = torch.randn(c, generator=g) * 0.1
            p[name + ".running_var"] = torch.rand(c, generator=g) * 0.5 + 0.75
        else:
            p[name + ".weight"], p[name + ".bias"] = torch.ones(c), torch.zeros(c)
            p[name + ".running_mean"], p[name + ".running_var"] = torch.zeros(c), torch.ones(c)

    bb = "backbone."
    for j, cin in enumerate((3, 64, 64)):
        msra("%sstem.conv%d.weight" % (bb, j + 1), 64, cin, 3)
        bn("%sstem.conv%d.norm" % (bb, j + 1), 64)
    p[bb + "stem.conv1.weight"] *= input_gain
    basic = depth in (18, 34)
    cin = 64
    for si, nblocks in enumerate(STAGES_WSR[depth]):
        stage = "res%d" % (si + 2)
        cout = (64 if basic else 256) * 2 ** si
        mid = 64 * 2 ** si
        for b in range(nblocks):
            pre = "%s%s.%d." % (bb, stage, b)
            if cin != cout:
                msra(pre + "shortcut.weight", cout, cin, 1)
                bn(pre + "shortcut.norm", cout)
            if basic:
                msra(pre + "conv1.weight", cout, cin, 3)
                bn(pre + "conv1.norm", cout)
                msra(pre + "conv2.weight", cout, cout, 3)
                bn(pre + "conv2.norm", cout)
            else:
                msra(pre + "conv1.weight", mid, cin, 1)
                bn(pre + "conv1.norm", mid)
                msra(pre + "conv2.weight", mid, mid, 3)
                bn(pre + "conv2.norm", mid)
                msra(pre + "conv3.weight", cout, mid, 1)
                bn(pre + "conv3.norm", cout)
            cin = cout
    c5 = cin
    rh = "roi_heads."
    d_in = c5 * 7 * 7
    for i, d in enumerate(dan_dims):
        p["%sbox_head.fc%d.weight" % (rh, i + 1)] = torch.randn(d, d_in, generator=g) * (0.005 if i else 0.002)
        p["%sbox_head.fc%d.bias" % (rh, i + 1)] = torch.full((d,), 0.1)
        d_in = d
    nmil = nt + ns - 1
    for nm in ("cls", "det"):
        bound = math.sqrt(6.0 / (d_in + nmil))
        p["%sbox_predictor.%s.weight" % (rh, nm)] = (torch.rand(nmil, d_in, generator=g) * 2 - 1) * bound
        p["%sbox_predictor.%s.bias" % (rh, nm)] = torch.zeros(nmil)
    for k in range(refine_k):
        p["%sbox_refinery_%d.cls_score.weight" % (rh, k)] = torch.randn(nt + 1, d_in, generator=g) * 0.01
        p["%sbox_refinery_%d.cls_score.bias" % (rh, k)] = torch.zeros(nt + 1)
        p["%sbox_refinery_%d.bbox_pred.weight" % (rh, k)] = torch.randn(nt * 4, d_in, generator=g) * 0.001
        p["%sbox_refinery_%d.bbox_pred.bias" % (rh, k)] = torch.zeros(nt * 4)
    for head in ("mask_head", "mask_refinery_0"):
        for k in range(4):
            msra("%s%s.mask_fcn%d.weight" % (rh, head, k + 1), 256, c5 if k == 0 else 256, 3)
            p["%s%s.mask_fcn%d.bias" % (rh, head, k + 1)] = torch.zeros(256)
        p["%s%s.deconv.weight" % (rh, head)] = torch.randn(256, 256, 2, 2, generator=g) * math.sqrt(2.0 / (256 * 4))
        p["%s%s.deconv.bias" % (rh, head)] = torch.zeros(256)
        p["%s%s.predictor.weight" % (rh, head)] = torch.randn(nt, 256, 1, 1, generator=g) * 0.001
        p["%s%s.predictor.bias" % (rh, head)] = torch.zeros(nt)
    p["pixel_mean"] = torch.tensor(PIXEL_MEAN).view(-1, 1, 1)
    p["pixel_std"] = torch.ones(3, 1, 1)
    return p


def wsr_v2_dc5(p, x, depth, res5_dilation=2):
    """ResNet-WS v2 forward (resnet_wsl_v2.py:102-119 BasicBlock, :229-251 BottleneckBlock, :418-429 stem, :694-700
    stage table): no strided convolution; res3 / res4 pool their first block's input."""
    bb = "backbone."
    for j in (1, 2, 3):
        x = conv_bn(p, "%sstem.conv%d" % (bb, j), x, 2 if j == 1 else 1, 1, True)
    x = F.max_pool2d(x, 2, 2)
    basic = depth in (18, 34)
    for si, nblocks in enumerate(STAGES_WSR[depth]):
        number = si + 2
        dil = res5_dilation if number in (4, 5) else 1
        for b in range(nblocks):
            pre = "%sres%d.%d." % (bb, number, b)
            if b == 0 and number in (3, 4):
                if number == 3 or res5_dilation == 1:
                    x = F.max_pool2d(x, 2, 2)
                else:
                    x = F.max_pool2d(F.pad(x, (0, 1, 0, 1)), 2, 1)
            sc = conv_bn(p, pre + "shortcut", x) if (pre + "shortcut.weight") in p else x
            if basic:
                out = conv_bn(p, pre + "conv1", x, 1, dil, True, dil)
                out = conv_bn(p, pre + "conv2", out, 1, dil, False, dil)
            else:
                out = conv_bn(p, pre + "conv1", x, 1, 0, True)
                out = conv_bn(p, pre + "conv2", out, 1, dil, True, dil)
                out = conv_bn(p, pre + "conv3", out)
            x = F.relu(out + sc)
    return x


# ----------------------------------------------------------------------------- box utilities
def pairwise_iou(a, b):
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    wh = (torch.min(a[:, None, 2:], b[:, 2:]) - torch.max(a[:, None, :2], b[:, :2])).clamp(min=0)
    inter = wh.prod(dim=2)
    return torch.where(inter > 0, inter / (area_a[:, None] + area_b - inter), torch.zeros(1, dtype=inter.dtype))


BBOX_W = (10.0, 10.0, 5.0, 5.0)
SCALE_CLAMP = math.log(1000.0 / 16)


def box_deltas(src, tgt):
    sw, sh = src[:, 2] - src[:, 0], src[:, 3] - src[:, 1]
    sx, sy = src[:, 0] + 0.5 * sw, src[:, 1] + 0.5 * sh
    tw, th = tgt[:, 2] - tgt[:, 0], tgt[:, 3] - tgt[:, 1]
    tx, ty = tgt[:, 0] + 0.5 * tw, tgt[:, 1] + 0.5 * th
    wx, wy, ww, wh = BBOX_W
    return torch.stack((wx * (tx - sx) / sw, wy * (ty - sy) / sh, ww * torch.log(tw / sw), wh * torch.log(th / sh)), 1)


def apply_deltas(deltas, boxes):
    """deltas (R, k*4), boxes (R,4) -> (R, k*4)."""
    boxes = boxes.to(deltas.dtype)
    w, h = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    cx, cy = boxes[:, 0] + 0.5 * w, boxes[:, 1] + 0.5 * h
    wx, wy, ww, wh = BBOX_W
    dx, dy = deltas[:, 0::4] / wx, deltas[:, 1::4] / wy
    dw = (deltas[:, 2::4] / ww).clamp(max=SCALE_CLAMP)
    dh = (deltas[:, 3::4] / wh).clamp(max=SCALE_CLAMP)
    pcx, pcy = dx * w[:, None] + cx[:, None], dy * h[:, None] + cy[:, None]
    pw, ph = torch.exp(dw) * w[:, None], torch.exp(dh) * h[:, None]
    out = torch.zeros_like(deltas)
    out[:, 0::4], out[:, 1::4] = pcx - 0.5 * pw, pcy - 0.5 * ph
    out[:, 2::4], out[:, 3::4] = pcx + 0.5 * pw, pcy + 0.5 * ph
    return out


def assign_levels(boxes, min_level=2, max_level=5, canonical_size=224, canonical_level=4):
    sizes = torch.sqrt((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]))
    lv = torch.floor(canonical_level + torch.log2(sizes / canonical_size + 1e-8))
    return torch.clamp(lv, min=min_level, max=max_level).to(torch.int64) - min_level


def match_and_label(prop_boxes, tgt, nt=NUM_THINGS):
    """IoU-match every proposal to the pseudo GT of its image (threshold 0.5, no sub-sampling)."""
    iou = pairwise_iou(tgt["boxes"], prop_boxes)
    if iou.shape[0] == 0:
        R = prop_boxes.shape[0]
        return dict(classes=torch.full((R,), nt, dtype=torch.int64), idx=torch.zeros(R, dtype=torch.int64))
    val, idx = iou.max(dim=0)
    classes = tgt["classes"][idx].clone()
    classes[val < 0.5] = nt
    return dict(classes=classes, idx=idx, boxes=tgt["boxes"][idx], scores=tgt["scores"][idx],
                weights=tgt["weights"][idx])


def mine_top1(boxes_per_class, scores, class_ids, img_probs):
    """One pseudo box per present class: the top-scoring proposal of that class column.
    boxes_per_class (R, ncls, 4); scores (R, >=ncls); class_ids (G,) int64; img_probs (ncls_total,)."""
    sc = scores[:, class_ids]                      # (R, G)
    top, idx = torch.topk(sc, 1, dim=0)            # (1, G)
    bx = boxes_per_class[:, class_ids]             # (R, G, 4)
    picked = torch.gather(bx, 0, idx[:, :, None].expand(1, class_ids.numel(), 4))[0]
    return dict(boxes=picked, classes=class_ids.clone(), scores=top[0], weights=img_probs[class_ids], idx=idx[0])


# ----------------------------------------------------------------------------- pooling glue
def rois_with_batch(boxes_list):
    return torch.cat([torch.cat([torch.full((len(b), 1), float(i), dtype=b.dtype), b], 1)
                      for i, b in enumerate(boxes_list)])


def moi_pool_levels(feats, boxes_list, oh_list, superpixels, res=7, single=False):
    """Multi-level MOIPool.  The reference only works single-level (F2); defined here as: each roi is
    pooled on its FPN level with that level's scale; output AND argmax are scattered back."""
    rois = rois_with_batch(boxes_list)
    L = max(o.shape[1] for o in oh_list)
    oh = torch.cat([F.pad(o, (0, L - o.shape[1])) for o in oh_list]).to(torch.int32)
    lv = torch.zeros(rois.shape[0], dtype=torch.int64) if single else assign_levels(rois[:, 1:])
    C = feats[0].shape[1]
    out = torch.zeros(rois.shape[0], C, res, res)
    arg = torch.full((rois.shape[0], C, res, res), -1, dtype=torch.int32)
    grads = []
    for l, f in enumerate(feats):
        sel = torch.nonzero(lv == l)[:, 0]
        if sel.numel() == 0:
            continue
        scale = f.shape[2] / superpixels.shape[1]
        y, a = P.moi_pool_forward(f.detach().numpy(), rois[sel].numpy(), scale, res, res, oh[sel].numpy(),
                                  superpixels.numpy())
        y, a = torch.from_numpy(y), torch.from_numpy(a)
        if _FORCED is not None and "argmax" in _FORCED:      # winners decided elsewhere: read the features there
            a = _FORCED["argmax"][sel].to(torch.int32)
            b_idx = rois[sel, 0].to(torch.int64)
            flat = f.detach().flatten(2)                                                 # (B, C, H*W)
            picked = torch.gather(flat[b_idx], 2, a.clamp(min=0).to(torch.int64).flatten(2)).view(a.shape)
            y = torch.where(a >= 0, picked, torch.zeros(()))
        out[sel], arg[sel] = y, a
        grads.append((l, sel, scale))
    return _MoiLevels.apply(out, arg, rois, grads, res, *feats), arg


class _MoiLevels(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out, arg, rois, plan, res, *feats):
        ctx.plan, ctx.res, ctx.shapes = plan, res, [f.shape for f in feats]
        ctx.save_for_backward(arg, rois)
        return out.clone()

    @staticmethod
    def backward(ctx, g):
        arg, rois = ctx.saved_tensors
        gs = [None] * len(ctx.shapes)
        for l, sel, scale in ctx.plan:
            B, C, H, W = ctx.shapes[l]
            gs[l] = torch.from_numpy(P.moi_pool_backward(
                np.ascontiguousarray(g[sel].numpy()), rois[sel].numpy(), arg[sel].numpy(), scale, ctx.res,
                ctx.res, B, C, H, W))
        gs = [torch.zeros(s) if x is None else x for x, s in zip(gs, ctx.shapes)]
        return (None, None, None, None, None, *gs)


class _RoiAlignLevel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f, rois, scale, res):
        ctx.cfg = (scale, res, f.shape)
        ctx.save_for_backward(rois)
        return torch.from_numpy(P.roi_align_forward(f.detach().numpy(), rois.numpy(), scale, res, res, 0, True))

    @staticmethod
    def backward(ctx, g):
        (rois,) = ctx.saved_tensors
        scale, res, (B, C, H, W) = ctx.cfg
        return torch.from_numpy(P.roi_align_backward(np.ascontiguousarray(g.numpy()), rois.numpy(), scale, res,
                                                     res, B, C, H, W, 0, True)), None, None, None


def roi_align_levels(feats, rois, img_size, res, single=False):
    """ROIPooler with ROIAlignV2 (aligned=True, sampling_ratio=0), detectron2/modeling/poolers.py:190-249
    (single: one feature map, no level assignment, :216-217)."""
    lv = torch.zeros(rois.shape[0], dtype=torch.int64) if single else assign_levels(rois[:, 1:])
    out = torch.zeros(rois.shape[0], feats[0].shape[1], res, res)
    for l, f in enumerate(feats):
        sel = torch.nonzero(lv == l)[:, 0]
        if sel.numel():
            out[sel] = _RoiAlignLevel.apply(f, rois[sel], f.shape[2] / img_size, res)
    return out


# ----------------------------------------------------------------------------- losses
def mil_scores(C, D, counts):
    return torch.cat([F.softmax(c, dim=1) * F.softmax(d, dim=0)
                      for c, d in zip(C.split(counts), D.split(counts))])


def mil_image_probs(scores, counts):
    return torch.cat([s.sum(0, keepdim=True) for s in scores.split(counts)]).clamp(1e-6, 1.0 - 1e-6)


def oicr_losses(logits, deltas, prop_boxes, lab, nt=NUM_THINGS):
    """Weighted CE / weighted L1 of one refinement branch; lab: dict of cat'ed classes, boxes, weights."""
    w = lab["weights"].clone()
    w[lab["classes"] == -1] = 0.0
    valid = (w > 1e-12).to(w.dtype).sum()
    ce = F.cross_entropy(logits, lab["classes"], reduction="none", ignore_index=-1)
    loss_cls = (ce * w).sum() / valid
    fg = torch.nonzero((lab["classes"] >= 0) & (lab["classes"] < nt))[:, 0]
    cols = 4 * lab["classes"][fg][:, None] + torch.arange(4)
    tgt = box_deltas(prop_boxes, lab["boxes"])
    l1 = (deltas[fg[:, None], cols] - tgt[fg]).abs()          # smooth_l1 with beta = 0
    loss_box = (l1 * w[fg, None]).sum() / lab["classes"].numel()
    return loss_cls, loss_box


# ----------------------------------------------------------------------------- heads
def linear_relu_drop(p, name, x, drop):
    x = gated_relu(F.linear(x, p[name + ".weight"], p[name + ".bias"]), name)
    return x if drop is None else x * drop


def image_labels(gt_classes_list, sem_seg, nt=NUM_THINGS, ns=NUM_STUFF):
    """(B,133) one-hot of present things ‖ present stuff (roi_heads.py:145-161; roi_heads_jtsm.py:165-194)."""
    things = [torch.unique(g) for g in gt_classes_list]
    stuff = []
    for s in sem_seg:
        u = torch.unique(s)
        u = u[(u != 255) & (u != 0)] - 1
        stuff.append(u.to(torch.int64))
    oh = torch.zeros(len(things), nt + ns - 1)
    for i, (t, s) in enumerate(zip(things, stuff)):
        oh[i, t] = 1
        oh[i, nt + s] = 1
    return things, [s + nt for s in stuff], oh


def eroded_rect_masks(boxes, H, W, erode=2):
    """(G,H,W) float masks of the boxes shrunk by `erode` px (stand-in for grabCut, F8)."""
    ys, xs = torch.arange(H).view(1, H, 1) + 0.5, torch.arange(W).view(1, 1, W) + 0.5
    b = boxes.view(-1, 4, 1, 1)
    return ((xs >= b[:, 0] + erode) & (xs <= b[:, 2] - erode) & (ys >= b[:, 1] + erode) &
            (ys <= b[:, 3] - erode)).to(torch.float32)


def evidence_masks(oh_rows, sp):
    """(G,H,W) bool superpixel-evidence masks (object_evidence, roi_heads_jtsm.py:1928-1994, superpixel branch):
    mask_g = union of the superpixels l < L with oh_rows[g, l] != 0 (the reference's `poses` loop runs over range(L),
    so a superpixel id outside [0, L) belongs to no mask)."""
    L = oh_rows.shape[1]
    ids = sp.reshape(1, -1).to(torch.int64)
    inside = (ids >= 0) & (ids < L)
    hit = torch.gather(oh_rows.to(torch.int64), 1, ids.clamp(0, L - 1).expand(oh_rows.shape[0], -1)) != 0
    return (hit & inside).reshape(oh_rows.shape[0], *sp.shape)


def pgt_sem_seg(tgt_list, H, W, nt=NUM_THINGS, oh_list=None, sp=None):
    """get_pgt_sem_seg (roi_heads_jtsm.py:2025-2070): the targets' masks painted with class - nt + 1 in ascending
    score order (:2061-2062), then every class absent from the map painted once more in list order (:2064-2066).
    Masks: the targets' pgt_masks = superpixel evidence of their oh_labels rows (oh_list / sp given; :2038-2047 with
    need_mask=True -> :1333-1334 -> :1928-1994); without them round 1's rectangles shrunk by 2 px."""
    out = torch.zeros(len(tgt_list), H, W, dtype=torch.int64)
    for i, t in enumerate(tgt_list):
        if oh_list is not None:
            masks = evidence_masks(oh_list[i][t["idx"]], sp[i])
        else:
            masks = eroded_rect_masks(t["boxes"], H, W) > 0.5
        vals = t["classes"] - nt + 1
        # (equal scores: the reference's argsort leaves their order open; fixed here and in the product as "lower
        # index first")
        for j in torch.argsort(t["scores"], descending=False, stable=True):
            out[i][masks[j]] = vals[j]
        for j in range(vals.numel()):
            if not (out[i] == vals[j]).any():
                out[i][masks[j]] = vals[j]
    return out


def mask_head_layers(p, pre, x):
    for k in range(4):
        x = gated_relu(F.conv2d(x, p["%smask_fcn%d.weight" % (pre, k + 1)], p["%smask_fcn%d.bias" % (pre, k + 1)], 1, 1),
                       "%smask_fcn%d" % (pre, k + 1))
    x = gated_relu(F.conv_transpose2d(x, p[pre + "deconv.weight"], p[pre + "deconv.bias"], 2), pre + "deconv")
    return F.conv2d(x, p[pre + "predictor.weight"], p[pre + "predictor.bias"])


def semseg_head(p, feats):
    """feats: dict p2..p5.  Returns logits at stride 4 (B,54,H/4,W/4)."""
    total = None
    for lvl, nconv in zip((2, 3, 4, 5), (1, 1, 2, 3)):
        x = feats["p%d" % lvl]
        for j in range(nconv):
            idx = j * (1 if lvl == 2 else 2)
            n = "sem_seg_head.p%d.%d" % (lvl, idx)
            x = gated_relu(F.group_norm(F.conv2d(x, p[n + ".weight"], None, 1, 1), 32, p[n + ".norm.weight"],
                                        p[n + ".norm.bias"]), n)
            if lvl != 2:
                x = F.interpolate(x, scale_factor=2.0, mode="bilinear", align_corners=False)
        total = x if total is None else total + x
    return F.conv2d(total, p["sem_seg_head.predictor.weight"], p["sem_seg_head.predictor.bias"])


# ----------------------------------------------------------------------------- the step
def near_targets_and_masks(boxes, sel, tgt, oh, sp, top_k=10):
    """Mask labels of one image.  boxes (R,4) proposals, sel (N,) foreground rows, tgt: top-1 pseudo GT, oh (R,L)
    int labels, sp (H,W) int superpixels.  Returns (near (G*k,) proposal rows, matched (N,) index into near,
    masks (G*k,H,W) float evidence masks)."""
    n = sel.numel()
    if n == 0 or tgt["boxes"].shape[0] == 0:
        return sel.new_zeros(0), sel.new_zeros(0), torch.zeros((0,) + tuple(sp.shape))
    iou = pairwise_iou(tgt["boxes"], boxes[sel])                              # (G, N) targets x fg proposals
    k = min(n, top_k)
    order = torch.argsort(iou, dim=1, descending=True, stable=True)[:, :k]    # ties: lower row first
    near = sel[order.reshape(-1)]                                             # pseudo box major, rank minor
    matched = pairwise_iou(boxes[near], boxes[sel]).max(dim=0).indices        # first maximum (matcher.py:61-103)
    masks = (torch.gather(oh[near].to(torch.int64), 1, sp.reshape(1, -1).to(torch.int64).expand(near.numel(), -1))
             != 0).to(torch.float32).reshape(near.numel(), *sp.shape)
    return near, matched, masks


def forward_losses(p, batch, depth=50, refine_k=4, dropout_masks=None, return_aux=False, arch="fpn",
                   nt=NUM_THINGS, ns=NUM_STUFF, mask_targets="evidence", forced=None, sem_targets="evidence"):
    """batch: dict(images=[(3,H,W)], boxes=[(R_i,4)], objectness=[(R_i,)], oh_labels=[(R_i,L) int],
    superpixels=(B,H,W) int32, gt_classes=[(n_i,) int64], sem_seg=(B,H,W) int64).
    arch "fpn": the R50/R101-FPN composite with SemSegFPNHead; arch "dc5": the shipped single-level composite
    (ResNet-WS v2 dilated C5, one feature map at stride 8, TwoClassHead = no semantic loss), nt thing / ns semantic
    classes.  Returns the loss dict (keys as in SURVEY §5 'Metrics / logging')."""
    global _FORCED
    _FORCED = forced
    try:
        return _forward_losses(p, batch, depth, refine_k, dropout_masks, return_aux, arch, nt, ns, mask_targets,
                               sem_targets)
    finally:
        _FORCED = None


def _forward_losses(p, batch, depth, refine_k, dropout_masks, return_aux, arch, nt, ns, mask_targets, sem_targets):
    single = arch == "dc5"
    NUM_THINGS, NUM_MIL = nt, nt + ns - 1   # (shadow the module constants: the body below is written with them)
    x = preprocess(p, batch["images"], 8 if single else 32)
    Himg, Wimg = x.shape[2:]
    if single:
        feats = {"res5": wsr_v2_dc5(p, x, depth)}
        levels = [feats["res5"]]
    else:
        feats = resnet_fpn(p, x, depth)
        levels = [feats["p%d" % l] for l in (2, 3, 4, 5)]
    counts = [len(b) for b in batch["boxes"]]
    things, stuff, labels_oh = image_labels(batch["gt_classes"], batch["sem_seg"], nt, ns)
    aux = {}

    # ---- box branch: MOIPool -> scale -> DAN -> MIL -> K OICR refinements
    pooled, argmax = moi_pool_levels(levels, batch["boxes"], batch["oh_labels"], batch["superpixels"], single=single)
    nvalid = (argmax[:, 0] != -1).reshape(argmax.shape[0], -1).sum(1).to(torch.float32)
    mask_scale = argmax.shape[2] * argmax.shape[3] * (nvalid + 1).reciprocal()
    pooled = pooled * mask_scale.view(-1, 1, 1, 1)
    obj = torch.cat([o + 1 for o in batch["objectness"]])
    pooled = pooled * obj.view(-1, 1, 1, 1)
    h = pooled.flatten(1)
    dm = dropout_masks or (None, None)
    h = linear_relu_drop(p, "roi_heads.box_head.fc1", h, dm[0])
    h = linear_relu_drop(p, "roi_heads.box_head.fc2", h, dm[1])
    C = F.linear(h, p["roi_heads.box_predictor.cls.weight"], p["roi_heads.box_predictor.cls.bias"])
    D = F.linear(h, p["roi_heads.box_predictor.det.weight"], p["roi_heads.box_predictor.det.bias"])
    scores = mil_scores(C, D, counts)
    probs = mil_image_probs(scores, counts)
    losses = {"loss_cls": F.binary_cross_entropy(probs, labels_oh, reduction="mean")}
    img_probs = probs.detach()
    aux.update(mil_scores=scores.detach(), img_probs=img_probs, pooled_argmax=argmax)

    all_boxes = torch.cat(batch["boxes"])
    prev_scores = list(scores.detach().split(counts))
    prev_boxes = [b[:, None, :].expand(len(b), NUM_MIL, 4) for b in batch["boxes"]]
    sem_tgts = [mine_top1(pb, ps, st, ip) for pb, ps, st, ip in zip(prev_boxes, prev_scores, stuff, img_probs)]
    sp_pad = F.pad(batch["superpixels"], (0, Wimg - batch["superpixels"].shape[2], 0, Himg - batch["superpixels"].shape[1]))
    if sem_targets == "evidence":
        sem_target = pgt_sem_seg(sem_tgts, Himg, Wimg, nt, batch["oh_labels"], sp_pad)
    else:
        sem_target = pgt_sem_seg(sem_tgts, Himg, Wimg, nt)
    prev_boxes = [pb[:, :NUM_THINGS] for pb in prev_boxes]
    for k in range(refine_k):
        tg = [mine_top1(pb, ps, th, ip) for pb, ps, th, ip in zip(prev_boxes, prev_scores, things, img_probs)]
        lab = [match_and_label(b, t, nt) for b, t in zip(batch["boxes"], tg)]
        cat = {f: torch.cat([l[f] for l in lab]) for f in ("classes", "boxes", "weights")}
        pre = "roi_heads.box_refinery_%d." % k
        z = F.linear(h, p[pre + "cls_score.weight"], p[pre + "cls_score.bias"])
        d = F.linear(h, p[pre + "bbox_pred.weight"], p[pre + "bbox_pred.bias"])
        losses["loss_cls_r%d" % k], losses["loss_box_reg_r%d" % k] = oicr_losses(z, d, all_boxes, cat, nt)
        prev_scores = list(F.softmax(z.detach(), dim=-1).split(counts))
        prev_boxes = [pb.view(-1, NUM_THINGS, 4) for pb in apply_deltas(d.detach(), all_boxes).split(counts)]
        aux["pgt_idx_r%d" % k] = [t["idx"] for t in tg]
        aux["labels_r%d" % k] = cat["classes"]

    # ---- mask branch: top-1 pseudo GT from the last refinement -> fg proposals -> two mask heads
    tg = [mine_top1(pb, ps, th, ip) for pb, ps, th, ip in zip(prev_boxes, prev_scores, things, img_probs)]
    lab = [match_and_label(b, t, nt) for b, t in zip(batch["boxes"], tg)]
    fg_rois, fg_cls, tgt_rois, gt_masks, near_rows = [], [], [], [], []
    base = row0 = 0
    for i, (b, l, t) in enumerate(zip(batch["boxes"], lab, tg)):
        sel = torch.nonzero((l["classes"] >= 0) & (l["classes"] < NUM_THINGS))[:, 0]
        fg_rois.append(torch.cat([torch.full((len(sel), 1), float(i)), b[sel]], 1))
        fg_cls.append(l["classes"][sel])
        if mask_targets == "evidence":
            near, matched, masks = near_targets_and_masks(b, sel, t, batch["oh_labels"][i], sp_pad[i])
            tgt_rois.append(torch.cat([(base + matched).to(b.dtype)[:, None], b[sel]], 1))
            gt_masks.append(masks)
            near_rows.append(near + row0)
            base += masks.shape[0]
        else:
            tgt_rois.append(torch.cat([(base + l["idx"][sel]).to(b.dtype)[:, None], b[sel]], 1))
            gt_masks.append(eroded_rect_masks(t["boxes"], Himg, Wimg))
            base += t["boxes"].shape[0]
        row0 += b.shape[0]
    fg_rois, fg_cls, tgt_rois = torch.cat(fg_rois), torch.cat(fg_cls), torch.cat(tgt_rois)
    aux.update(fg_rois=fg_rois, fg_classes=fg_cls, near_rows=near_rows)
    mfeat = roi_align_levels(levels, fg_rois, Himg, 14, single=single)
    gt28 = torch.from_numpy(P.roi_align_forward(torch.cat(gt_masks)[:, None].numpy(), tgt_rois.numpy(), 1.0, 28,
                                                28, 0, True))[:, 0] >= 0.5
    ar = torch.arange(fg_rois.shape[0])
    logits = mask_head_layers(p, "roi_heads.mask_head.", mfeat)
    def mask_loss(z, target):   # mask_rcnn_loss: an empty foreground set contributes an exact zero (mask_head.py:47-48)
        if z.shape[0] == 0:
            return z.sum() * 0
        return F.binary_cross_entropy_with_logits(z[ar, fg_cls], target.to(torch.float32))

    losses["loss_mask"] = mask_loss(logits, gt28)
    aux["mask_targets"] = gt28
    if mask_targets == "evidence" and fg_rois.shape[0] > 0:
        # get_pgt_mask: the class probability pasted into the image at the proposal box, cropped back to the box
        from .inference import paste_masks_in_image
        prob = logits.detach()[ar, fg_cls].sigmoid()
        tgt2 = torch.zeros_like(gt28)
        for s0 in range(0, fg_rois.shape[0], 64):   # (chunks: the pasted images are full size)
            sl_ = slice(s0, min(s0 + 64, fg_rois.shape[0]))
            pasted = paste_masks_in_image(prob[sl_], fg_rois[sl_, 1:], (Himg, Wimg), 0.5).to(torch.float32)
            rr = torch.cat([torch.arange(pasted.shape[0], dtype=torch.float32)[:, None], fg_rois[sl_, 1:]], 1)
            tgt2[sl_] = torch.from_numpy(P.roi_align_forward(pasted[:, None].numpy(), rr.numpy(), 1.0, 28, 28, 0,
                                                             True))[:, 0] >= 0.5
    else:
        tgt2 = logits.detach()[ar, fg_cls] > 0.0      # sigmoid > 0.5 of the first head's own prediction
    if _FORCED is not None and "mask_targets_r0" in _FORCED:
        tgt2 = _FORCED["mask_targets_r0"]
    aux["mask_targets_r0"] = tgt2
    logits2 = mask_head_layers(p, "roi_heads.mask_refinery_0.", mfeat)
    losses["loss_mask_r0"] = mask_loss(logits2, tgt2)

    # ---- semantic branch (the TwoClassHead of the shipped configs has no loss, seg_heads.py:231-275)
    if not single:
        sl = semseg_head(p, feats)
        sl = F.interpolate(sl, scale_factor=4.0, mode="bilinear", align_corners=False)
        losses["loss_sem_seg"] = F.cross_entropy(sl, sem_target, reduction="mean", ignore_index=255)
    aux["sem_target"] = sem_target
    return (losses, aux) if return_aux else losses


# ----------------------------------------------------------------------------- synthetic batch
def synthetic_batch(seed, B=2, size=1024, R=2000, sp_block=32, n_things=3, n_stuff=2, nt=NUM_THINGS, ns=NUM_STUFF,
                    cluster=0.0, objects=None):
    """cluster > 0: that fraction of every image's proposals are jittered copies (each edge +-12 % of the side) of
    `objects` rectangles per image — bench.py's headline recipe (jtsm_amd/utils/synthetic.py), where a mined pseudo box
    has many proposals above IoU 0.5 and crowded, nearly equal IoUs (drawn from a second generator, so cluster = 0
    reproduces the plain recipe bit for bit)."""
    g = torch.Generator().manual_seed(seed)
    g2 = torch.Generator().manual_seed(seed * 7919 + 13)
    grid = size // sp_block
    ids = (torch.arange(size)[:, None] // sp_block) * grid + (torch.arange(size)[None, :] // sp_block)
    cy = (torch.arange(grid) * sp_block + sp_block / 2.0)
    out = dict(images=[], boxes=[], objectness=[], oh_labels=[], gt_classes=[],
               superpixels=ids.to(torch.int32)[None].repeat(B, 1, 1))
    sem = torch.zeros(B, size, size, dtype=torch.int64)
    for i in range(B):
        out["images"].append(torch.rand(3, size, size, generator=g) * 255)
        x0 = torch.rand(R, generator=g) * size * 0.75
        y0 = torch.rand(R, generator=g) * size * 0.75
        lo, hi = math.log(16.0), math.log(size / 2.0)
        w = torch.exp(torch.rand(R, generator=g) * (hi - lo) + lo)
        hh = torch.exp(torch.rand(R, generator=g) * (hi - lo) + lo)
        bx = torch.stack([x0, y0, (x0 + w).clamp(max=size), (y0 + hh).clamp(max=size)], 1)
        if cluster > 0:
            nc, no = int(round(cluster * R)), (n_things if objects is None else int(objects))
            ox, oy = torch.rand(no, generator=g2) * size * 0.6, torch.rand(no, generator=g2) * size * 0.6
            ow = torch.exp(torch.rand(no, generator=g2) * (hi - lo) + lo)
            ohh = torch.exp(torch.rand(no, generator=g2) * (hi - lo) + lo)
            which = torch.randint(0, no, (nc,), generator=g2)
            jit = (torch.rand(nc, 4, generator=g2) - 0.5) * 0.24
            cl = torch.stack([ox, oy, ox + ow, oy + ohh], 1)[which] + jit * torch.stack([ow, ohh, ow, ohh], 1)[which]
            bx[:nc] = cl.clamp(0, size)
        out["boxes"].append(bx)
        out["objectness"].append(torch.rand(R, generator=g))
        iny = (cy[None, :] >= bx[:, 1:2]) & (cy[None, :] <= bx[:, 3:4])
        inx = (cy[None, :] >= bx[:, 0:1]) & (cy[None, :] <= bx[:, 2:3])
        out["oh_labels"].append((iny[:, :, None] & inx[:, None, :]).reshape(R, -1).to(torch.int32))
        out["gt_classes"].append(torch.randperm(nt, generator=g)[:n_things].sort().values)
        st = torch.randperm(ns - 1, generator=g)[:n_stuff] + 1
        band = size // (n_stuff + 1)
        for j, s in enumerate(st):
            sem[i, (j + 1) * band:(j + 2) * band] = s
        sem[i, :8] = 255
    out["sem_seg"] = sem
    return out
