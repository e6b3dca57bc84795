"""CPU suite: the inference / post-processing oracle (oracle/inference.py) against independent readings of the
same definitions — the reference's tests for this row need torchvision / pycocotools / COCO files and hold no golden
vectors (tests/layers/test_nms.py, test_mask_ops.py), so the restatement is pinned by properties and by literal
brute-force loops."""
import numpy as np
import torch

from oracle import inference as OI


def _random_boxes(n, size, g):
    xy = torch.rand(n, 2, generator=g) * size * 0.8
    wh = torch.rand(n, 2, generator=g) * size * 0.3 + 1.0
    return torch.cat([xy, xy + wh], dim=1)


def _iou(a, b):
    iw = max(min(a[2], b[2]) - max(a[0], b[0]), 0.0)
    ih = max(min(a[3], b[3]) - max(a[1], b[1]), 0.0)
    inter = iw * ih
    return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter)


def test_nms_matches_literal_loop():
    g = torch.Generator().manual_seed(0)
    boxes, scores = _random_boxes(300, 200, g), torch.rand(300, generator=g)
    for thr in (0.2, 0.5, 0.8):     # the thresholds of the reference's tests/layers/test_nms.py:27
        keep = OI.nms(boxes, scores, thr).tolist()
        order = sorted(range(300), key=lambda i: (-float(scores[i]), i))
        want = []
        for i in order:
            if all(_iou(boxes[k].tolist(), boxes[i].tolist()) <= thr for k in want):
                want.append(i)
        assert keep == want


def test_batched_nms_is_per_class():
    g = torch.Generator().manual_seed(1)
    boxes, scores = _random_boxes(400, 200, g), torch.rand(400, generator=g)
    idxs = torch.randint(0, 7, (400,), generator=g)
    keep = OI.batched_nms(boxes, scores, idxs, 0.5)
    assert torch.equal(scores[keep], torch.sort(scores[keep], descending=True).values)
    for c in range(7):
        m = torch.nonzero(idxs == c).view(-1)
        per = m[OI.nms(boxes[m], scores[m], 0.5)]
        assert sorted(per.tolist()) == sorted(k for k in keep.tolist() if idxs[k] == c)


def test_fast_rcnn_inference_drops_nonfinite_and_thresholds():
    g = torch.Generator().manual_seed(2)
    R, K = 50, 5
    boxes = _random_boxes(R * K, 100, g).view(R, K * 4)
    scores = torch.softmax(torch.randn(R, K + 1, generator=g) * 3, dim=1)
    scores[3, 1] = float("nan")
    boxes[7, 2] = float("inf")
    det = OI.fast_rcnn_inference_single_image(boxes, scores, (80, 90), 0.05, 0.5, 20)
    assert len(det["boxes"]) <= 20 and 3 not in det["rows"].tolist() and 7 not in det["rows"].tolist()
    assert (det["scores"] > 0.05).all() and (det["classes"] < K).all()
    assert (det["boxes"][:, 2] <= 90).all() and (det["boxes"][:, 3] <= 80).all() and (det["boxes"] >= 0).all()
    assert torch.equal(det["scores"], scores[det["rows"], det["classes"]])


def test_paste_of_constant_mask_is_the_box():
    masks = torch.ones(2, 28, 28)
    boxes = torch.tensor([[10.0, 20.0, 50.0, 70.0], [0.0, 0.0, 31.5, 16.25]])
    out = OI.paste_masks_in_image(masks, boxes, (96, 80))
    assert out.dtype == torch.bool and out.shape == (2, 96, 80)
    assert out[0, 21:69, 11:49].all() and not out[0, :19].any() and not out[0, :, 52:].any()
    assert out[1, :15, :30].all() and not out[1, 18:].any()


def test_combine_hand_example():
    H = W = 8
    sem = torch.zeros(H, W, dtype=torch.int64)
    sem[4:] = 2
    sem[:, 6:] = 3
    masks = torch.zeros(3, H, W, dtype=torch.bool)
    masks[0, :4, :4] = True          # accepted (best score)
    masks[1, :4, :3] = True          # inside instance 0: overlap 1.0 > 0.5 -> skipped
    masks[2, 2:6, 2:6] = True        # overlaps 4 of 16 pixels: accepted, only the free 12 painted
    scores = torch.tensor([0.9, 0.8, 0.7])
    classes = torch.tensor([5, 6, 7])
    pan, info = OI.combine_semantic_and_instance_outputs(masks, scores, classes, sem, 0.5, 4, 0.5)
    assert [s["category_id"] for s in info if s["isthing"]] == [5, 7]
    assert (pan[:4, :4] == 1).all() and int((pan == 2).sum()) == 12
    stuff = [s for s in info if not s["isthing"]]
    assert [s["category_id"] for s in stuff] == [2, 3]
    assert stuff[0]["area"] == int(((sem == 2) & (pan != 1) & (pan != 2)).sum())
    # confidence threshold stops the walk
    pan2, info2 = OI.combine_semantic_and_instance_outputs(masks, scores, classes, sem, 0.5, 4, 0.85)
    assert [s["category_id"] for s in info2 if s["isthing"]] == [5]


def test_predict_K_is_mean_of_heads():
    g = torch.Generator().manual_seed(3)
    zs = [torch.randn(6, 4, generator=g) for _ in range(3)]
    ds = [torch.randn(6, 12, generator=g) * 0.1 for _ in range(3)]
    prop = _random_boxes(6, 50, g)
    probs, boxes = OI.predict_K(zs, ds, prop)
    assert torch.allclose(probs.sum(1), torch.ones(6), atol=1e-6)
    assert torch.allclose(probs, sum(torch.softmax(z, 1) for z in zs) / 3, atol=1e-7)
    assert boxes.shape == (6, 12)
    zero = OI.predict_K(zs, [torch.zeros(6, 12)] * 2, prop)[1]
    assert torch.allclose(zero, prop.repeat(1, 3), atol=1e-4)


def test_nms_matches_the_reference_greedy_loop():
    """tests/golden/nms_ref.npz: keep lists produced by the REFERENCE's own compiled greedy NMS
    (detectron2/layers/csrc/nms_rotated/nms_rotated_cpu.cpp, "modified from torchvision's nms_cpu_kernel") on
    axis-aligned boxes, every pairwise IoU at least 1e-3 away from the threshold — pins the oracle's visiting order
    and suppress rule to the reference run (generator: tests/golden/make_golden.py)."""
    from conftest import load_cases

    cases = load_cases("nms_ref.npz")
    assert len(cases) == 4
    for name, c in cases.items():
        keep = OI.nms(torch.from_numpy(c["boxes"]), torch.from_numpy(c["scores"]), float(c["thr"][0]))
        assert keep.tolist() == c["keep"].tolist(), name
        one_class = OI.batched_nms(torch.from_numpy(c["boxes"]), torch.from_numpy(c["scores"]),
                                   torch.zeros(len(c["boxes"]), dtype=torch.int64), float(c["thr"][0]))
        assert one_class.tolist() == c["keep"].tolist(), name
