"""Fused pseudo-GT mining / labelling (jtsm_amd/csrc/mining.hip): device-side replacements for the
get_pgt_top_k + label_and_sample_proposals glue of JTSMROIHeads (no autograd: label generation)."""
import torch

from .. import _lib as L


def pad_class_lists(class_lists, device):
    """list of per-image int64 class-id tensors -> (classes (B,Gmax) int32, counts (B,) int32, Gmax)."""
    gmax = max(1, max(int(c.numel()) for c in class_lists))
    cls = torch.zeros((len(class_lists), gmax), dtype=torch.int32, device=device)
    for i, c in enumerate(class_lists):
        cls[i, : c.numel()] = c.to(torch.int32)
    counts = torch.tensor([int(c.numel()) for c in class_lists], dtype=torch.int32).to(device, non_blocking=True)
    return cls, counts, gmax


@torch.no_grad()
def row_lse(logits):
    L.require_gpu(logits)
    assert logits.dim() == 2 and logits.stride(1) == 1
    out = torch.empty(logits.shape[0], dtype=torch.float32, device=logits.device)
    L.check(L.lib().jtsm_row_lse_f32(L.ptr(logits), logits.stride(0), logits.shape[1], logits.shape[0], L.ptr(out),
                                     L.stream()), "row_lse")
    return out


@torch.no_grad()
def mine_top1(scores, proposals, bag_offsets, classes, counts, img_probs, lse=None, deltas=None):
    """-> dict(idx (B,G) int32, boxes (B,G,4), scores (B,G), weights (B,G))."""
    L.require_gpu(scores, proposals)
    assert scores.stride(1) == 1 and (deltas is None or deltas.stride(1) == 1)
    B, G = classes.shape
    dev = scores.device
    # one zero-filled allocation (one fill launch, not four) carved into the four outputs: 7 words per (image, class)
    words = torch.zeros((7, B, G), dtype=torch.int32, device=dev)
    out = dict(idx=words[0], boxes=words[1:5].view(torch.float32).view(B, G, 4), scores=words[5].view(torch.float32),
               weights=words[6].view(torch.float32))
    proposals = proposals.contiguous()
    img_probs = img_probs.contiguous()
    L.check(L.lib().jtsm_mine_top1_f32(
        L.ptr(scores), scores.stride(0), L.ptr(lse), L.ptr(proposals), L.ptr(deltas),
        deltas.stride(0) if deltas is not None else 0, L.ptr(bag_offsets), L.ptr(classes), L.ptr(counts), B, G,
        L.ptr(img_probs), img_probs.shape[1], L.ptr(out["idx"]), L.ptr(out["boxes"]), L.ptr(out["scores"]),
        L.ptr(out["weights"]), L.stream()), "mine_top1")
    return out


@torch.no_grad()
def match_label(proposals, bag_offsets, pgt, classes, counts, bg_label, iou_thresh=0.5):
    """-> dict(labels (R,) int32, matched (R,) int32, boxes (R,4), weights (R,), scores (R,))."""
    proposals = proposals.contiguous()
    R, dev = proposals.shape[0], proposals.device
    B, G = classes.shape
    out = dict(labels=torch.empty(R, dtype=torch.int32, device=dev), matched=torch.empty(R, dtype=torch.int32, device=dev),
               boxes=torch.empty((R, 4), dtype=torch.float32, device=dev),
               weights=torch.empty(R, dtype=torch.float32, device=dev),
               scores=torch.empty(R, dtype=torch.float32, device=dev))
    L.check(L.lib().jtsm_match_label_f32(
        L.ptr(proposals), L.ptr(bag_offsets), B, R, L.ptr(pgt["boxes"]), L.ptr(classes), L.ptr(counts),
        L.ptr(pgt["weights"]), L.ptr(pgt["scores"]), G, L.f32(iou_thresh), int(bg_label), L.ptr(out["labels"]),
        L.ptr(out["matched"]), L.ptr(out["boxes"]), L.ptr(out["weights"]), L.ptr(out["scores"]), L.stream()),
        "match_label")
    return out


@torch.no_grad()
def paint_sem_seg(boxes, classes, scores, counts, class_base, height, width, erode=2.0):
    """Pseudo semantic target (B,H,W) int64 from per-image pseudo boxes (B,G,4) / classes (B,G) / scores (B,G) and
    their counts (B,): rectangles shrunk by `erode`, value classes - class_base, best score on top, then classes
    left without a pixel painted once more in list order (libjtsm_hip.so: jtsm_paint_sem_seg)."""
    L.require_gpu(boxes, scores)
    B, G = classes.shape
    boxes, scores = boxes.contiguous(), scores.contiguous()
    classes, counts = classes.to(torch.int32).contiguous(), counts.to(torch.int32).contiguous()
    lib = L.lib()
    out = torch.empty((B, height, width), dtype=torch.int64, device=boxes.device)
    ws = torch.empty(lib.jtsm_paint_sem_seg_workspace_bytes(B), dtype=torch.uint8, device=boxes.device)
    L.check(lib.jtsm_paint_sem_seg(L.ptr(boxes), L.ptr(classes), L.ptr(scores), L.ptr(counts), B, G, int(class_base),
                                   height, width, L.f32(erode), L.ptr(out), L.ptr(ws), L.stream()), "paint_sem_seg")
    return out


@torch.no_grad()
def paint_sem_seg_evidence(target_idx, bag_offsets, oh_labels, superpixels, classes, scores, counts, class_base):
    """Pseudo semantic target (B,H,W) int64 painted from the targets' superpixel-evidence masks, as the reference does
    (roi_heads_jtsm.py:2038-2069): target j of image b is proposal row bag_offsets[b] + target_idx[b,j]; its mask is the
    union of the superpixels that row of oh_labels (R,L) marks.  Ascending-score paint order, then classes left without
    a pixel once more in list order (libjtsm_hip.so: jtsm_paint_sem_seg_evidence)."""
    L.require_gpu(oh_labels, superpixels, scores)
    B, G = classes.shape
    _, h, w = superpixels.shape
    assert superpixels.shape[0] == B and oh_labels.dim() == 2
    oh_labels = oh_labels.to(torch.int32).contiguous()
    superpixels = superpixels.to(torch.int32).contiguous()
    classes, counts = classes.to(torch.int32).contiguous(), counts.to(torch.int32).contiguous()
    lib = L.lib()
    out = torch.empty((B, h, w), dtype=torch.int64, device=scores.device)
    ws = torch.empty(lib.jtsm_paint_sem_seg_workspace_bytes(B), dtype=torch.uint8, device=scores.device)
    L.note_bytes(12.0 * B * h * w)   # superpixel ids read, int64 target written
    L.check(lib.jtsm_paint_sem_seg_evidence(
        L.ptr(target_idx.to(torch.int32).contiguous()), L.ptr(bag_offsets), L.ptr(oh_labels), oh_labels.shape[1],
        L.ptr(superpixels), L.ptr(classes), L.ptr(scores.contiguous()), L.ptr(counts), B, G, int(class_base), h, w,
        L.ptr(out), L.ptr(ws), L.stream()), "paint_sem_seg_evidence")
    return out


@torch.no_grad()
def rect_mask_targets(rois, rects, side, height, width, erode=2.0):
    """(N, side, side) bool mask targets: the H x W bitmask of rects[n] shrunk by `erode`, cropped and resized to
    rois[n] the way BitMasks.crop_and_resize does it (ROIAlign, aligned, adaptive sampling, >= 0.5) — evaluated
    analytically, no bitmask is rasterised (libjtsm_hip.so: jtsm_rect_mask_targets_f32)."""
    L.require_gpu(rois, rects)
    rois, rects = rois.to(torch.float32).contiguous(), rects.to(torch.float32).contiguous()
    n = rois.shape[0]
    out = torch.empty((n, side, side), dtype=torch.uint8, device=rois.device)
    L.check(L.lib().jtsm_rect_mask_targets_f32(L.ptr(rois), L.ptr(rects), L.ptr(out), n, side, height, width,
                                               L.f32(erode), L.stream()), "rect_mask_targets")
    return out.view(torch.bool)      # (0 / 1 bytes: the same memory read as bool, no pass)


@torch.no_grad()
def near_targets(proposals, bag_offsets, labels, bg_label, pgt, counts, top_k=10):
    """The mask branch's "top_k nearest" targets (roi_heads_jtsm.py:840-905), sync-free:
    -> (near_rows (B,G,top_k) int32, matched_near (R,) int32); see jtsm_near_targets_f32."""
    proposals = proposals.contiguous()
    R, dev = proposals.shape[0], proposals.device
    B, G = pgt["boxes"].shape[:2]
    near = torch.empty((B, G, top_k), dtype=torch.int32, device=dev)
    matched = torch.empty(R, dtype=torch.int32, device=dev)
    L.check(L.lib().jtsm_near_targets_f32(L.ptr(proposals), L.ptr(bag_offsets), B, R, L.ptr(labels), int(bg_label),
                                          L.ptr(pgt["boxes"]), L.ptr(counts), G, int(top_k), L.ptr(near),
                                          L.ptr(matched), L.stream()), "near_targets")
    return near, matched


@torch.no_grad()
def sp_mask_targets(rois, oh_rows, img_of, oh_labels, superpixels, side):
    """(N, side, side) bool targets from superpixel evidence: the union of the superpixels marked by row oh_rows[n]
    of oh_labels, cropped to rois[n] (libjtsm_hip.so: jtsm_sp_mask_targets_f32)."""
    L.require_gpu(rois, oh_labels, superpixels)
    rois = rois.to(torch.float32).contiguous()
    oh_labels = oh_labels.to(torch.int32).contiguous()
    superpixels = superpixels.to(torch.int32).contiguous()
    n = rois.shape[0]
    _, h, w = superpixels.shape
    out = torch.empty((n, side, side), dtype=torch.uint8, device=rois.device)
    L.check(L.lib().jtsm_sp_mask_targets_f32(L.ptr(rois), L.ptr(oh_rows.to(torch.int32).contiguous()),
                                             L.ptr(img_of.to(torch.int32).contiguous()), L.ptr(oh_labels),
                                             oh_labels.shape[1], L.ptr(superpixels), L.ptr(out), n, side, h, w,
                                             L.stream()), "sp_mask_targets")
    return out.view(torch.bool)      # (0 / 1 bytes: the same memory read as bool, no pass)


@torch.no_grad()
def paste_crop_targets(probs, rois, side, height, width, threshold=0.5):
    """(N, side, side) bool targets of the mask refinery: probs (N, M, M) pasted at rois and cropped back
    (libjtsm_hip.so: jtsm_paste_crop_targets_f32)."""
    L.require_gpu(probs, rois)
    probs, rois = probs.to(torch.float32).contiguous(), rois.to(torch.float32).contiguous()
    n, m = probs.shape[0], probs.shape[-1]
    out = torch.empty((n, side, side), dtype=torch.uint8, device=rois.device)
    L.check(L.lib().jtsm_paste_crop_targets_f32(L.ptr(probs), L.ptr(rois), L.ptr(out), n, m, side, height, width,
                                                L.f32(threshold), L.stream()), "paste_crop_targets")
    return out.view(torch.bool)      # (0 / 1 bytes: the same memory read as bool, no pass)


# ---- label preparation (csrc/mining.hip, round 3): per-image inputs travel as pointers, no concatenation ----------
MAX_IMAGES = 16


def _image_rows(tensors, dtype):
    """(pointer array, count array) of per-image device tensors for the `B pointers + counts` entry points; the
    tensors are made contiguous / cast here and returned so that they outlive the launch."""
    import ctypes as C
    keep = [t.to(dtype).contiguous() for t in tensors]
    ptrs = (C.c_void_p * len(keep))(*[t.data_ptr() if t.numel() else None for t in keep])
    counts = (C.c_int * len(keep))(*[int(t.shape[0]) for t in keep])
    return ptrs, counts, keep


@torch.no_grad()
def pooler_rois_levels(box_tensors, min_level, max_level, canonical_box_size, canonical_level):
    """list of per-image (n_i, 4) float32 boxes -> (rois (M,5) float32, level (M,) int32): detectron2's
    convert_boxes_to_pooler_format + assign_boxes_to_levels in one launch, the same levels bit for bit."""
    import ctypes as C
    L.require_gpu(*box_tensors)
    ptrs, counts, keep = _image_rows([b.reshape(-1, 4) for b in box_tensors], torch.float32)
    M, dev = sum(int(t.shape[0]) for t in keep), keep[0].device
    rois = torch.empty((M, 5), dtype=torch.float32, device=dev)
    level = torch.empty(M, dtype=torch.int32, device=dev)
    L.check(L.lib().jtsm_pooler_rois_levels_f32(ptrs, counts, len(keep), int(min_level), int(max_level),
                                                C.c_float(canonical_box_size), C.c_float(canonical_level), L.ptr(rois),
                                                L.ptr(level), L.stream()), "pooler_rois_levels")
    return rois, level


@torch.no_grad()
def roi_scale(argmax, objectness_list):
    """bins / (valid bins + 1) * (objectness + 1) per roi; argmax (M,C,P,P) int32 channels-last (MOIPool's)."""
    L.require_gpu(argmax)
    M, Cc, ph, pw = argmax.shape
    assert argmax.dtype == torch.int32 and argmax.is_contiguous(memory_format=torch.channels_last)
    ptrs, counts, keep = _image_rows(objectness_list, torch.float32)
    assert sum(int(t.shape[0]) for t in keep) == M
    out = torch.empty(M, dtype=torch.float32, device=argmax.device)
    L.check(L.lib().jtsm_roi_scale_f32(L.ptr(argmax), ph * pw, Cc, ptrs, counts, len(keep), L.ptr(out), L.stream()),
            "roi_scale")
    return out


@torch.no_grad()
def image_labels(gt_classes_list, num_classes, gt_sem_seg=None, num_stuff=0, stuff_offset=0):
    """-> (oh_things (B,C) float32, things_cls (B,C) int32, things_cnt (B,) int32, oh_stuff, stuff_cls, stuff_cnt);
    the stuff triple is None without `gt_sem_seg` ((B,H,W) int64 or uint8)."""
    L.require_gpu(*gt_classes_list)
    B, dev = len(gt_classes_list), gt_classes_list[0].device
    ptrs, counts, keep = _image_rows(gt_classes_list, torch.int64)
    import ctypes as C

    def triple(width):      # one allocation carved into (presence (B,width) float32, list (B,width) int32, count (B,) int32)
        words = torch.empty(B * (2 * width + 1), dtype=torch.int32, device=dev)
        return (words[:B * width].view(torch.float32).view(B, width), words[B * width:2 * B * width].view(B, width),
                words[2 * B * width:])

    oh_t, cls_t, cnt_t = triple(num_classes)
    oh_s = cls_s = cnt_s = sem = ws = None
    esz = pixels = 0
    if gt_sem_seg is not None:
        sem = gt_sem_seg if gt_sem_seg.dtype in (torch.int64, torch.uint8) else gt_sem_seg.to(torch.int64)
        sem = sem.contiguous()
        assert sem.shape[0] == B
        esz, pixels = sem.element_size(), sem[0].numel()
        oh_s, cls_s, cnt_s = triple(num_stuff - 1)
        ws = torch.empty(L.lib().jtsm_image_labels_workspace_bytes(B), dtype=torch.uint8, device=dev)
    L.check(L.lib().jtsm_image_labels(ptrs, counts, B, num_classes, L.ptr(sem), esz, C.c_long(pixels), num_stuff,
                                      stuff_offset, L.ptr(oh_t), L.ptr(cls_t), L.ptr(cnt_t), L.ptr(oh_s), L.ptr(cls_s),
                                      L.ptr(cnt_s), L.ptr(ws), L.stream()), "image_labels")
    return oh_t, cls_t, cnt_t, oh_s, cls_s, cnt_s


@torch.no_grad()
def fg_compact(labels, bg_label, bag_offsets, boxes, matched=None):
    """The foreground rows (labels != bg_label) in row order with what the mask branch reads of them, in ONE launch:
    -> dict(rows (R,) int32, boxes (R,4), classes (R,) int64, img (R,) int32, matched (R,) int32 | None,
    rois (R,5) float32, counts (B,) int64); of the R-long buffers the first counts.sum() entries are valid (slice
    them once the counts are on the host)."""
    L.require_gpu(labels, boxes)
    assert labels.dtype == torch.int32 and bag_offsets.dtype == torch.int32
    boxes = boxes.to(torch.float32).contiguous()
    R, B, dev = labels.shape[0], bag_offsets.numel() - 1, labels.device
    out = dict(rows=torch.empty(R, dtype=torch.int32, device=dev), boxes=torch.empty((R, 4), dtype=torch.float32, device=dev),
               classes=torch.empty(R, dtype=torch.int64, device=dev), img=torch.empty(R, dtype=torch.int32, device=dev),
               matched=torch.empty(R, dtype=torch.int32, device=dev) if matched is not None else None,
               rois=torch.empty((R, 5), dtype=torch.float32, device=dev),
               counts=torch.empty(B, dtype=torch.int64, device=dev))
    L.check(L.lib().jtsm_fg_compact(L.ptr(labels.contiguous()), int(bg_label), L.ptr(bag_offsets), B, R, L.ptr(boxes),
                                    L.ptr(matched), L.ptr(out["rows"]), L.ptr(out["boxes"]), L.ptr(out["classes"]),
                                    L.ptr(out["img"]), L.ptr(out["matched"]), L.ptr(out["rois"]), L.ptr(out["counts"]),
                                    L.stream()), "fg_compact")
    return out
