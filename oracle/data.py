"""ORACLE — TEST INFRASTRUCTURE ONLY.

Plain-numpy restatement of the input pipeline's geometry, written as literal per-element loops so
that it shares no code with jtsm_amd/data:
  shortest_edge_size     detectron2/data/transforms/augmentation_impl.py:152-172
  resize_box / flip_box  ResizeTransform.apply_coords (transform.py:143-146) / fvcore HFlipTransform.apply_coords
                         through Transform.apply_box's four corners (fvcore 0.1.2+, published behaviour)
  unique_rows            detectron2/structures/boxes.py:226-238 of the reference tree
  proposals_seg          projects/WSL/wsl/data/detection_utils.py:266-345
Parity pins: the reference's tests/data/test_transforms.py holds no golden vector for these ops
(it compares apply_image paths with each other): **parity unpinned**; pinned here by the literal loops."""
import numpy as np


def shortest_edge_size(h, w, size, max_size):
    scale = size * 1.0 / min(h, w)
    newh, neww = (size, scale * w) if h < w else (scale * h, size)
    if max(newh, neww) > max_size:
        s = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * s, neww * s
    return int(newh + 0.5), int(neww + 0.5)


def resize_box(box, h, w, new_h, new_w):
    xs = [box[0] * (new_w * 1.0 / w), box[2] * (new_w * 1.0 / w)]
    ys = [box[1] * (new_h * 1.0 / h), box[3] * (new_h * 1.0 / h)]
    return [min(xs), min(ys), max(xs), max(ys)]


def flip_box(box, width):
    xs = [width - box[0], width - box[2]]
    return [min(xs), box[1], max(xs), box[3]]


def unique_rows(boxes):
    seen, keep = set(), []
    for i, b in enumerate(boxes):
        key = int(round(float(b[0])) + round(float(b[1])) * 1e3 + round(float(b[2])) * 1e6 + round(float(b[3])) * 1e9)
        if key not in seen:
            seen.add(key)
            keep.append(i)
    return keep


def proposals_seg(boxes, scores, oh_labels, h, w, new_h, new_w, flip, topk, min_box_size=0):
    """Boxes (float) through resize (+ flip) -> clip -> unique -> non-empty -> top-k; returns (boxes, scores, oh)."""
    out = []
    for b in boxes:
        b = resize_box([float(v) for v in b], h, w, new_h, new_w)
        if flip:
            b = flip_box(b, new_w)
        b = np.asarray(b, dtype=np.float32)
        b = [min(max(b[0], 0), new_w), min(max(b[1], 0), new_h), min(max(b[2], 0), new_w), min(max(b[3], 0), new_h)]
        out.append(b)
    out = np.asarray(out, dtype=np.float32).reshape(-1, 4)
    keep = unique_rows(out)
    keep = [i for i in keep if (out[i, 2] - out[i, 0]) > min_box_size and (out[i, 3] - out[i, 1]) > min_box_size]
    keep = keep[:topk]
    return out[keep], np.asarray(scores, dtype=np.float32)[keep], np.asarray(oh_labels)[keep]
