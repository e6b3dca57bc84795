"""Inference / post-processing operators (jtsm_amd/csrc/postprocess.hip) — SURVEY §8f row 4.

Thin ctypes wrappers: allocate outputs and workspaces, call libjtsm_hip.so on the current stream.  None of the
library calls synchronises; a wrapper that returns a data-dependent number of rows reads one int32 from the device
at its very end (the reference's `nonzero()` / `.item()` calls synchronise at the same places)."""
import ctypes as C

import torch

from .. import _lib as L


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def _ptr_array(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _sizet(fn, *args):
    fn.restype = C.c_size_t
    return int(fn(*args))


@torch.no_grad()
def oicr_predict(logits_heads, deltas_heads, proposal_boxes, weights, scale_clamp):
    """Mean over heads of the row soft-max and Box2BoxTransform.apply_deltas of the mean deltas
    (fast_rcnn_oicr.py:712-783).  -> probs (R, K+1), boxes (R, 4*Kb) or None."""
    logits_heads = [z.contiguous() for z in logits_heads]
    L.require_gpu(*logits_heads)
    R, C1 = logits_heads[0].shape
    dev = logits_heads[0].device
    probs = torch.empty((R, C1), dtype=torch.float32, device=dev)
    boxes = dptr = None
    kb = 0
    if deltas_heads is not None:
        deltas_heads = [d.contiguous() for d in deltas_heads]
        L.require_gpu(*deltas_heads)
        kb = deltas_heads[0].shape[1] // 4
        boxes = torch.empty((R, kb * 4), dtype=torch.float32, device=dev)
        dptr = _ptr_array(deltas_heads)
        proposal_boxes = proposal_boxes.contiguous()
    w = (C.c_float * 4)(*[float(v) for v in weights]) if deltas_heads is not None else None
    L.check(L.lib().jtsm_oicr_predict_f32(_ptr_array(logits_heads), dptr, len(logits_heads), R, C1, kb,
                                          L.ptr(proposal_boxes) if boxes is not None else None, w, L.f32(scale_clamp),
                                          L.ptr(probs), L.ptr(boxes), L.stream()), "oicr_predict")
    return probs, boxes


@torch.no_grad()
def batched_nms_device(boxes, scores, idxs, iou_threshold, num_classes, max_per_class, coordinate_trick=2):
    """-> (keep (n,) int64 survivors first, num_keep (1,) int32, overflow (1,) int32), all on the device."""
    L.require_gpu(boxes, scores, idxs)
    n = boxes.shape[0]
    dev = boxes.device
    keep = torch.empty(n, dtype=torch.int64, device=dev)
    num = torch.zeros(1, dtype=torch.int32, device=dev)
    ovf = torch.zeros(1, dtype=torch.int32, device=dev)
    if n == 0:
        return keep, num, ovf
    boxes = boxes.to(torch.float32).contiguous()
    scores = scores.to(torch.float32).contiguous()
    idxs = idxs.to(torch.int64).contiguous()
    lib = L.lib()
    nbytes = _sizet(lib.jtsm_batched_nms_workspace_bytes, n, int(num_classes), int(max_per_class))
    ws = _ws(nbytes, dev)
    L.check(lib.jtsm_batched_nms_f32(L.ptr(boxes), L.ptr(scores), L.ptr(idxs), n, int(num_classes), int(max_per_class),
                                     L.f32(iou_threshold), int(coordinate_trick), L.ptr(keep), L.ptr(num), L.ptr(ovf),
                                     L.ptr(ws), C.c_size_t(ws.numel()), L.stream()), "batched_nms")
    return keep, num, ovf


@torch.no_grad()
def fast_rcnn_inference_device(boxes, scores, image_shape, score_thresh, nms_thresh, topk):
    """One image of fast_rcnn_inference_single_image on the device.  boxes (R, 4*Kb), scores (R, K+1).
    -> dict(boxes (cap,4), scores, classes, rows, count (1,) int32); cap = topk, or R*K when topk < 0."""
    L.require_gpu(boxes, scores)
    boxes, scores = boxes.to(torch.float32).contiguous(), scores.to(torch.float32).contiguous()
    R, K = scores.shape[0], scores.shape[1] - 1
    kb = boxes.shape[1] // 4
    dev = boxes.device
    cap = int(topk) if topk >= 0 else R * K
    out = dict(boxes=torch.empty((cap, 4), dtype=torch.float32, device=dev),
               scores=torch.empty(cap, dtype=torch.float32, device=dev),
               classes=torch.empty(cap, dtype=torch.int64, device=dev),
               rows=torch.empty(cap, dtype=torch.int64, device=dev),
               count=torch.zeros(1, dtype=torch.int32, device=dev))
    lib = L.lib()
    ws = _ws(_sizet(lib.jtsm_fast_rcnn_inference_workspace_bytes, R, K), dev)
    L.check(lib.jtsm_fast_rcnn_inference_f32(
        L.ptr(boxes), L.ptr(scores), R, K, kb, L.f32(image_shape[0]), L.f32(image_shape[1]), L.f32(score_thresh),
        L.f32(nms_thresh), int(topk), cap, L.ptr(out["boxes"]), L.ptr(out["scores"]), L.ptr(out["classes"]),
        L.ptr(out["rows"]), L.ptr(out["count"]), L.ptr(ws), C.c_size_t(ws.numel()), L.stream()), "fast_rcnn_inference")
    return out


@torch.no_grad()
def mask_probs(logits_heads, classes):
    """sigmoid(mean over heads of the predicted class's channel): (N, C, M, M) x heads -> (N, 1, M, M)."""
    logits_heads = [z.contiguous() for z in logits_heads]
    L.require_gpu(*logits_heads)
    N, Cc, M, M2 = logits_heads[0].shape
    assert M == M2
    out = torch.empty((N, 1, M, M), dtype=torch.float32, device=logits_heads[0].device)
    classes = classes.to(torch.int64).contiguous()
    L.check(L.lib().jtsm_mask_probs_f32(_ptr_array(logits_heads), len(logits_heads), L.ptr(classes), N, Cc, M,
                                        L.ptr(out), L.stream()), "mask_probs")
    return out


@torch.no_grad()
def paste_masks(masks, boxes, img_h, img_w, threshold=0.5):
    """(N, M, M) soft masks -> (N, img_h, img_w) bool (threshold >= 0) / uint8 (threshold < 0)."""
    L.require_gpu(masks, boxes)
    masks, boxes = masks.to(torch.float32).contiguous(), boxes.to(torch.float32).contiguous()
    N, M = masks.shape[0], masks.shape[-1]
    out = torch.empty((N, int(img_h), int(img_w)), dtype=torch.uint8, device=masks.device)
    for s in range(0, N, 65535):
        e = min(N, s + 65535)
        L.check(L.lib().jtsm_paste_masks_f32(L.ptr(masks[s:e]), L.ptr(boxes[s:e]), e - s, M, int(img_h), int(img_w),
                                             L.f32(threshold), L.ptr(out[s:e]), L.stream()), "paste_masks")
    return out.view(torch.bool) if threshold >= 0 else out


@torch.no_grad()
def resize_bilinear(x, out_hw, crop_hw=None, scale_factor=None):
    """F.interpolate(x[..., :crop_h, :crop_w], bilinear, align_corners=False) -> planar (N, C, oh, ow) float32.
    x: (N, C, H, W), plain or channels_last storage."""
    L.require_gpu(x)
    N, Cc, H, W = x.shape
    ch, cw = (H, W) if crop_hw is None else (int(crop_hw[0]), int(crop_hw[1]))
    oh, ow = int(out_hw[0]), int(out_hw[1])
    nhwc = L.is_nhwc(x)
    if not nhwc:
        x = x.contiguous()
    if scale_factor is not None:
        sh = sw = 1.0 / float(scale_factor)
    else:   # area_pixel_compute_scale: float(in) / out, computed in float32
        sh = float(torch.tensor(ch, dtype=torch.float32) / torch.tensor(oh, dtype=torch.float32)) if oh > 0 else 0.0
        sw = float(torch.tensor(cw, dtype=torch.float32) / torch.tensor(ow, dtype=torch.float32)) if ow > 0 else 0.0
    y = torch.empty((N, Cc, oh, ow), dtype=torch.float32, device=x.device)
    L.check(L.lib().jtsm_resize_bilinear_f32(L.ptr(x), L.NHWC if nhwc else L.NCHW, N, Cc, H, W, ch, cw, oh, ow,
                                             L.f32(sh), L.f32(sw), L.ptr(y), L.stream()), "resize_bilinear")
    return y


@torch.no_grad()
def argmax_channels(x):
    """(C, H, W) planar float32 -> (H, W) int64, first maximum wins."""
    L.require_gpu(x)
    x = x.contiguous()
    Cc, H, W = x.shape
    out = torch.empty((H, W), dtype=torch.int64, device=x.device)
    L.check(L.lib().jtsm_argmax_channels_f32(L.ptr(x), Cc, C.c_long(H * W), L.ptr(out), L.stream()), "argmax_channels")
    return out


@torch.no_grad()
def panoptic_combine(masks, scores, classes, sem, num_sem_classes, overlap_threshold, stuff_area_limit,
                     instances_confidence_threshold):
    """-> panoptic (H, W) int32, seg_table (n, 5) int32 rows {id, isthing, category_id, instance_id, area},
    seg_score (n,) — n read from the device once, at the end."""
    L.require_gpu(sem)
    H, W = sem.shape
    dev = sem.device
    N = 0 if masks is None else masks.shape[0]
    S = int(num_sem_classes)
    sem = sem.to(torch.int64).contiguous()
    pan = torch.empty((H, W), dtype=torch.int32, device=dev)
    table = torch.zeros((N + S, 5), dtype=torch.int32, device=dev)
    tscore = torch.zeros(N + S, dtype=torch.float32, device=dev)
    nseg = torch.zeros(1, dtype=torch.int32, device=dev)
    order = None
    if N:
        masks = masks.view(torch.uint8) if masks.dtype == torch.bool else masks.to(torch.uint8)
        masks = masks.contiguous()
        scores = scores.to(torch.float32).contiguous()
        classes = classes.to(torch.int64).contiguous()
        order = torch.sort(scores, descending=True, stable=True).indices.to(torch.int32)
        # instances above the confidence threshold: the walk is two launches per visited instance, so knowing the
        # count up front (one small read; this function ends with a read anyway) saves the rest
        visits = int((scores >= float(instances_confidence_threshold)).sum().item())
    lib = L.lib()
    ws = _ws(_sizet(lib.jtsm_panoptic_combine_workspace_bytes, N, S), dev)
    L.check(lib.jtsm_panoptic_combine(L.ptr(masks) if N else None, L.ptr(order), L.ptr(scores) if N else None,
                                      L.ptr(classes) if N else None, N, H, W, L.ptr(sem), S,
                                      C.c_double(float(overlap_threshold)), int(stuff_area_limit),
                                      L.f32(instances_confidence_threshold), L.ptr(pan), L.ptr(table), L.ptr(tscore),
                                      L.ptr(nseg), visits if N else -1, L.ptr(ws), C.c_size_t(ws.numel()), L.stream()),
            "panoptic_combine")
    n = int(nseg.item())
    return pan, table[:n], tscore[:n]


@torch.no_grad()
def preprocess_images_u8(images, mean, std, size_divisibility=0, pad_value=0.0):
    """uint8 (or float32) (C, h, w) device images -> ((B, C, Hp, Wp) float32 in channels-last storage, image_sizes):
    (x - mean) / std, zero padding, one launch (mcnn.py:303-318 + ImageList.from_tensors)."""
    L.require_gpu(*images)
    images = [im.contiguous() for im in images]
    if any(im.dtype != images[0].dtype for im in images) or images[0].dtype not in (torch.uint8, torch.float32):
        raise TypeError("preprocess_images: uint8 or float32 planes, one dtype per batch")
    entry = L.lib().jtsm_preprocess_images_u8 if images[0].dtype == torch.uint8 else L.lib().jtsm_preprocess_images_f32
    Cc = images[0].shape[0]
    sizes = [(int(im.shape[-2]), int(im.shape[-1])) for im in images]
    hp, wp = max(s[0] for s in sizes), max(s[1] for s in sizes)
    if size_divisibility > 1:
        hp = (hp + size_divisibility - 1) // size_divisibility * size_divisibility
        wp = (wp + size_divisibility - 1) // size_divisibility * size_divisibility
    B = len(images)
    out = torch.empty((B, Cc, hp, wp), dtype=torch.float32, device=images[0].device, memory_format=torch.channels_last)
    hs = (C.c_int32 * B)(*[s[0] for s in sizes])
    ws = (C.c_int32 * B)(*[s[1] for s in sizes])
    m = (C.c_float * Cc)(*[float(v) for v in mean])
    sd = (C.c_float * Cc)(*[float(v) for v in std])
    L.check(entry(_ptr_array(images), hs, ws, B, Cc, m, sd, L.f32(pad_value), hp, wp, L.ptr(out), L.stream()),
            "preprocess_images")
    return out, sizes


preprocess_images = preprocess_images_u8   # (either source dtype)


@torch.no_grad()
def resize_nearest(x, out_hw, flip_source=False):
    """F.interpolate(x[None], size=out_hw, mode="nearest")[0] of a planar (C, H, W) float map (source columns
    optionally mirrored first)."""
    L.require_gpu(x)
    x = x.to(torch.float32).contiguous()
    Cc, H, W = x.shape
    y = torch.empty((Cc, int(out_hw[0]), int(out_hw[1])), dtype=torch.float32, device=x.device)
    L.check(L.lib().jtsm_resize_nearest_f32(L.ptr(x), Cc, H, W, int(out_hw[0]), int(out_hw[1]), int(bool(flip_source)),
                                            L.ptr(y), L.stream()), "resize_nearest")
    return y
