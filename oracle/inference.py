"""ORACLE — TEST INFRASTRUCTURE ONLY.

torch-CPU / numpy restatement of the reference's inference + post-processing path.  It is the
checker for jtsm_amd/csrc/postprocess.hip; nothing in the product imports it.

What each function follows (paths relative to the reference tree):
  nms                         torchvision 0.8.1 (pinned by docker/Dockerfile:25; absent from /root/reference)
                              ops/csrc/cpu/nms_cpu.cpp — published algorithm: visit boxes by descending score; a
                              box is kept unless an earlier KEPT box has inter / (area_i + area_j - inter) > thr
  batched_nms                 detectron2/layers/nms.py:10-31 + torchvision 0.8.1 ops/boxes.py batched_nms
                              (boxes + idxs * (boxes.max() + 1)) below 40000 boxes, per-class loop above
  predict_K                   projects/WSL/wsl/modeling/roi_heads/fast_rcnn_oicr.py:712-783
  fast_rcnn_inference_single_image   .../fast_rcnn_oicr.py:100-163
  mask_rcnn_inference         projects/WSL/wsl/modeling/roi_heads/mask_head.py:106-147 on roi_heads_jtsm.py:949-961
  paste_masks_in_image        detectron2/layers/mask_ops.py:17-145 (GPU branch: whole image, F.grid_sample)
  detector_postprocess / sem_seg_postprocess   detectron2/modeling/postprocessing.py:10-100
  combine_semantic_and_instance_outputs        detectron2/modeling/meta_arch/panoptic_fpn.py:133-218
  forward_inference           projects/WSL/wsl/modeling/meta_arch/mcnn.py:236-365; roi_heads_jtsm.py:553-561,738-752

Parity pins: the greedy NMS loop is pinned to a REFERENCE RUN — tests/golden/nms_ref.npz holds keep lists of the
reference's own compiled detectron2/layers/csrc/nms_rotated/nms_rotated_cpu.cpp ("modified from torchvision's
nms_cpu_kernel") on angle-0 boxes whose pairwise IoUs all stay >= 1e-3 away from the threshold (generator:
tests/golden/make_golden.py through oracle/build_ref.py).  The reference's Python tests for this row need
torchvision / pycocotools / COCO files (tests/layers/test_nms.py, test_mask_ops.py) and hold no golden vectors:
**parity unpinned** for the ordering under equal scores (resolved here and in the product as "lower index first"; the
reference's sort is unstable there), for the coordinate-offset arithmetic of torchvision's batched_nms (restated from
its published source) and for the panoptic merge (hand-worked example).  grid_sample / interpolate / softmax are the
literal torch operators the reference calls.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import model as M


# ----------------------------------------------------------------------------- NMS
def nms(boxes, scores, thr):
    """Greedy NMS; returns kept indices by descending score (equal scores: ascending index).  float32 arithmetic."""
    b = boxes.detach().to(torch.float32).numpy()
    order = torch.sort(scores.detach().to(torch.float32), descending=True, stable=True).indices.numpy()
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    suppressed = np.zeros(len(b), dtype=bool)
    keep = []
    thr = np.float32(thr)
    for _i, i in enumerate(order):
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        xx1, yy1 = np.maximum(x1[i], x1[rest]), np.maximum(y1[i], y1[rest])
        xx2, yy2 = np.minimum(x2[i], x2[rest]), np.minimum(y2[i], y2[rest])
        w, h = np.maximum(np.float32(0), xx2 - xx1), np.maximum(np.float32(0), yy2 - yy1)
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr > thr]] = True
    return torch.as_tensor(np.asarray(keep, dtype=np.int64))


def batched_nms(boxes, scores, idxs, thr):
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    boxes = boxes.to(torch.float32)
    if len(boxes) < 40000:
        max_coordinate = boxes.max()
        offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
        return nms(boxes + offsets[:, None], scores, thr)
    result = torch.zeros(len(scores), dtype=torch.bool)
    for c in torch.unique(idxs).tolist():
        m = torch.nonzero(idxs == c).view(-1)
        result[m[nms(boxes[m], scores[m], thr)]] = True
    keep = torch.nonzero(result).view(-1)
    return keep[torch.sort(scores[keep], descending=True, stable=True).indices]


# ----------------------------------------------------------------------------- detections
def predict_K(logits_heads, deltas_heads, proposal_boxes):
    probs = torch.zeros_like(logits_heads[0])
    for z in logits_heads:
        probs += F.softmax(z, dim=-1)
    probs = probs / len(logits_heads)
    deltas = torch.zeros_like(deltas_heads[0])
    for d in deltas_heads:
        deltas += d
    deltas = deltas / len(deltas_heads)
    return probs, M.apply_deltas(deltas, proposal_boxes)


def fast_rcnn_inference_single_image(boxes, scores, image_shape, score_thresh, nms_thresh, topk):
    """-> dict(boxes, scores, classes, rows): rows index the INPUT proposals."""
    rows = torch.arange(scores.size(0))[:, None].repeat(1, scores.size(1))
    valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(scores).all(dim=1)
    boxes, scores, rows = boxes[valid], scores[valid], rows[valid]
    scores = scores[:, :-1]
    rows = rows[:, :-1]
    nreg = boxes.shape[1] // 4
    b = boxes.reshape(-1, 4).clone()
    h, w = image_shape
    b[:, 0].clamp_(min=0, max=w); b[:, 1].clamp_(min=0, max=h); b[:, 2].clamp_(min=0, max=w); b[:, 3].clamp_(min=0, max=h)
    b = b.view(-1, nreg, 4)
    filter_mask = scores > score_thresh
    filter_inds = filter_mask.nonzero()
    b = b[filter_inds[:, 0], 0] if nreg == 1 else b[filter_mask]
    scores, rows = scores[filter_mask], rows[filter_mask]
    keep = batched_nms(b, scores, filter_inds[:, 1], nms_thresh)
    if topk >= 0:
        keep = keep[:topk]
    return dict(boxes=b[keep], scores=scores[keep], classes=filter_inds[keep, 1], rows=rows[keep])


# ----------------------------------------------------------------------------- masks
def mask_rcnn_inference(logits_heads, classes):
    total = None
    for z in logits_heads:
        total = z.clone() if total is None else total + z
    z = total / len(logits_heads)
    if z.shape[1] == 1:
        return z.sigmoid()
    return z[torch.arange(z.shape[0]), classes][:, None].sigmoid()


def paste_masks_soft(masks, boxes, img_h, img_w):
    """_do_paste_mask(skip_empty=False): (N, M, M) -> (N, img_h, img_w) float."""
    masks = masks[:, None].to(torch.float32)
    x0, y0, x1, y1 = torch.split(boxes.to(torch.float32), 1, dim=1)
    N = masks.shape[0]
    img_y = torch.arange(0, img_h, dtype=torch.float32) + 0.5
    img_x = torch.arange(0, img_w, dtype=torch.float32) + 0.5
    img_y = (img_y - y0) / (y1 - y0) * 2 - 1
    img_x = (img_x - x0) / (x1 - x0) * 2 - 1
    gx = img_x[:, None, :].expand(N, img_y.size(1), img_x.size(1))
    gy = img_y[:, :, None].expand(N, img_y.size(1), img_x.size(1))
    grid = torch.stack([gx, gy], dim=3)
    return F.grid_sample(masks, grid, align_corners=False)[:, 0]


def paste_masks_in_image(masks, boxes, image_shape, threshold=0.5):
    if len(masks) == 0:
        return torch.zeros((0,) + tuple(image_shape), dtype=torch.bool)
    soft = paste_masks_soft(masks, boxes, int(image_shape[0]), int(image_shape[1]))
    return soft >= threshold if threshold >= 0 else (soft * 255).to(torch.uint8)


def detector_postprocess(det, image_size, out_h, out_w, mask_threshold=0.5):
    """det: dict(boxes, scores, classes[, masks (N,1,M,M)]) at `image_size` -> the same at (out_h, out_w)."""
    sx, sy = out_w / image_size[1], out_h / image_size[0]
    b = det["boxes"].clone()
    b[:, 0::2] *= sx
    b[:, 1::2] *= sy
    b[:, 0].clamp_(min=0, max=out_w); b[:, 1].clamp_(min=0, max=out_h); b[:, 2].clamp_(min=0, max=out_w); b[:, 3].clamp_(min=0, max=out_h)
    keep = ((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0)
    out = {k: v[keep] for k, v in det.items()}
    out["boxes"] = b[keep]
    if "masks" in out:
        out["soft_masks"] = paste_masks_soft(out["masks"][:, 0], out["boxes"], out_h, out_w) if keep.any() else \
            torch.zeros((0, out_h, out_w))
        out["masks"] = out["soft_masks"] >= mask_threshold
    return out


def sem_seg_postprocess(result, img_size, out_h, out_w):
    result = result[:, : img_size[0], : img_size[1]].expand(1, -1, -1, -1)
    return F.interpolate(result, size=(out_h, out_w), mode="bilinear", align_corners=False)[0]


# ----------------------------------------------------------------------------- panoptic merge
def combine_semantic_and_instance_outputs(masks, scores, classes, semantic_results, overlap_threshold, stuff_area_limit,
                                          instances_confidence_threshold):
    panoptic_seg = torch.zeros_like(semantic_results, dtype=torch.int32)
    sorted_inds = torch.sort(scores, descending=True, stable=True).indices if len(scores) else []
    current_segment_id = 0
    segments_info = []
    instance_masks = masks.to(torch.bool) if len(scores) else masks
    for inst_id in sorted_inds:
        score = scores[inst_id].item()
        if score < instances_confidence_threshold:
            break
        mask = instance_masks[inst_id]
        mask_area = mask.sum().item()
        if mask_area == 0:
            continue
        intersect = (mask > 0) & (panoptic_seg > 0)
        intersect_area = intersect.sum().item()
        if intersect_area * 1.0 / mask_area > overlap_threshold:
            continue
        if intersect_area > 0:
            mask = mask & (panoptic_seg == 0)
        current_segment_id += 1
        panoptic_seg[mask] = current_segment_id
        segments_info.append({"id": current_segment_id, "isthing": True, "score": score,
                              "category_id": classes[inst_id].item(), "instance_id": inst_id.item()})
    for semantic_label in torch.unique(semantic_results).tolist():
        if semantic_label == 0:
            continue
        mask = (semantic_results == semantic_label) & (panoptic_seg == 0)
        mask_area = mask.sum().item()
        if mask_area < stuff_area_limit:
            continue
        current_segment_id += 1
        panoptic_seg[mask] = current_segment_id
        segments_info.append({"id": current_segment_id, "isthing": False, "category_id": semantic_label,
                              "area": mask_area})
    return panoptic_seg, segments_info


# ----------------------------------------------------------------------------- whole inference pass
@torch.no_grad()
def forward_inference(p, batch, depth=50, refine_k=4, score_thresh=0.05, nms_thresh=0.5, topk=100,
                      overlap_threshold=0.5, stuff_area_limit=4096, instances_confidence_threshold=0.5,
                      return_raw=False):
    """batch as oracle.model.forward_losses takes (labels unused).  Returns one dict per image with `instances`
    (boxes, scores, classes, rows, masks), `sem_seg` (54, H, W) and `panoptic_seg` (map, segments_info)."""
    x = M.preprocess(p, batch["images"])
    feats = M.resnet_fpn(p, x, depth)
    levels = [feats["p%d" % l] for l in (2, 3, 4, 5)]
    counts = [len(b) for b in batch["boxes"]]
    pooled, argmax = M.moi_pool_levels(levels, batch["boxes"], batch["oh_labels"], batch["superpixels"])
    nvalid = (argmax[:, 0] != -1).reshape(argmax.shape[0], -1).sum(1).to(torch.float32)
    pooled = pooled * (argmax.shape[2] * argmax.shape[3] * (nvalid + 1).reciprocal()).view(-1, 1, 1, 1)
    pooled = pooled * torch.cat([o + 1 for o in batch["objectness"]]).view(-1, 1, 1, 1)
    h = pooled.flatten(1)
    h = M.linear_relu_drop(p, "roi_heads.box_head.fc1", h, None)
    h = M.linear_relu_drop(p, "roi_heads.box_head.fc2", h, None)
    zs, ds = [], []
    for k in range(refine_k):
        pre = "roi_heads.box_refinery_%d." % k
        zs.append(F.linear(h, p[pre + "cls_score.weight"], p[pre + "cls_score.bias"]))
        ds.append(F.linear(h, p[pre + "bbox_pred.weight"], p[pre + "bbox_pred.bias"]))
    probs, boxes = predict_K(zs, ds, torch.cat(batch["boxes"]))
    sl = M.semseg_head(p, feats)
    sl = F.interpolate(sl, scale_factor=4.0, mode="bilinear", align_corners=False)
    out = []
    dets = []
    for i, (pb, bb) in enumerate(zip(probs.split(counts), boxes.split(counts))):
        size = tuple(batch["images"][i].shape[-2:])
        dets.append((fast_rcnn_inference_single_image(bb, pb, size, score_thresh, nms_thresh, topk), size))
    rois = torch.cat([torch.cat([torch.full((len(d["boxes"]), 1), float(i)), d["boxes"]], 1)
                      for i, (d, _) in enumerate(dets)])
    mfeat = M.roi_align_levels(levels, rois, x.shape[2], 14)
    logits = M.mask_head_layers(p, "roi_heads.mask_refinery_0.", mfeat)
    mprob = mask_rcnn_inference([logits], torch.cat([d["classes"] for d, _ in dets]))
    if return_raw:
        return dict(probs=probs, boxes=boxes, dets=[d for d, _ in dets], mask_probs=mprob, sem_logits=sl)
    for (d, size), mp, s in zip(dets, mprob.split([len(d["boxes"]) for d, _ in dets]), sl):
        d = dict(d, masks=mp)
        inst = detector_postprocess(d, size, size[0], size[1])
        sem = sem_seg_postprocess(s, size, size[0], size[1])
        pan = combine_semantic_and_instance_outputs(inst["masks"], inst["scores"], inst["classes"], sem.argmax(dim=0),
                                                    overlap_threshold, stuff_area_limit,
                                                    instances_confidence_threshold)
        out.append({"instances": inst, "sem_seg": sem, "panoptic_seg": pan})
    return out
