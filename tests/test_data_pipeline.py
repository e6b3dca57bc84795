"""Input pipeline.  CPU: transforms / proposal files / mapper against oracle/data.py's literal
loops.  GPU: the fused normalise + pad + channels-last launch is bit-exact against the reference's expression, the
prefetcher hands over identical tensors and the model trains on them."""
import os

import numpy as np
import pytest
import torch

from jtsm_amd import data as D
from jtsm_amd.structures import Boxes, Instances
from oracle import data as OD


def _boxes(n, h, w, rng):
    xy = rng.rand(n, 2) * [w * 0.7, h * 0.7]
    wh = rng.rand(n, 2) * [w * 0.3, h * 0.3] + 1
    return np.concatenate([xy, xy + wh], 1)


def test_resize_shortest_edge_sizes():
    for (h, w, size, mx) in [(480, 640, 800, 1333), (1000, 300, 800, 1333), (333, 500, 1200, 4000), (600, 600, 608, 600)]:
        t = D.ResizeShortestEdge((size, size), mx).get_transform(np.zeros((h, w, 3), np.uint8))
        assert (t.new_h, t.new_w) == OD.shortest_edge_size(h, w, size, mx)


def test_box_transforms_match_literal_loops():
    rng = np.random.RandomState(0)
    b = _boxes(50, 375, 500, rng)
    rs = D.ResizeTransform(375, 500, 600, 800)
    fl = D.HFlipTransform(800)
    got = D.TransformList([rs, fl]).apply_box(b.copy())
    want = [OD.flip_box(OD.resize_box(list(x), 375, 500, 600, 800), 800) for x in b]
    assert np.allclose(got, np.asarray(want), rtol=0, atol=1e-9)
    back = D.TransformList([rs, fl]).inverse().apply_box(got.copy())
    assert np.allclose(back, b, atol=1e-9)


def test_image_and_segmentation_transforms():
    rng = np.random.RandomState(1)
    img = rng.randint(0, 256, (60, 90, 3)).astype(np.uint8)
    seg = rng.randint(0, 40, (60, 90)).astype(np.float32)
    t = D.TransformList([D.ResizeTransform(60, 90, 120, 180), D.HFlipTransform(180)])
    out = t.apply_image(img)
    assert out.shape == (120, 180, 3) and out.dtype == np.uint8
    s = t.apply_segmentation(seg)
    assert s.shape == (120, 180)
    # nearest resize by 2 then flip: every output pixel is the source pixel of its 2x2 cell, mirrored
    assert np.array_equal(s, seg.repeat(2, 0).repeat(2, 1)[:, ::-1])
    assert np.array_equal(D.HFlipTransform(90).apply_image(img), img[:, ::-1])


def _write_case(tmp_path, rng, h=120, w=160, R=80, L=20):
    boxes = _boxes(R, h, w, rng).astype(np.float32)
    boxes[5] = boxes[4]                                   # duplicate -> removed by unique_boxes
    boxes[7, 2] = boxes[7, 0]                             # empty -> removed by nonempty
    path = os.path.join(str(tmp_path), "p.pkl")
    D.write_proposal_file(path, boxes=boxes, scores=rng.rand(R, 1).astype(np.float32),
                          oh_labels=(rng.rand(R, L) > 0.5).astype(np.uint8),
                          superpixels=rng.randint(0, L, (h, w)).astype(np.int32), image_id=17)
    return path, boxes


def test_transform_proposals_seg_matches_oracle(tmp_path):
    rng = np.random.RandomState(2)
    path, boxes = _write_case(tmp_path, rng)
    raw = D.read_proposal_file(path)
    for flip in (False, True):
        tf = [D.ResizeTransform(120, 160, 240, 320)] + ([D.HFlipTransform(320)] if flip else [])
        d = {"proposal_file": path, "image_id": 17}
        D.transform_proposals_seg(d, (240, 320), D.TransformList(tf), proposal_topk=50)
        wb, ws, wo = OD.proposals_seg(raw["boxes"], np.squeeze(raw["scores"]), raw["oh_labels"], 120, 160, 240, 320,
                                      flip, 50)
        p = d["proposals"]
        assert len(p) == len(wb) <= 50 and len(wb) < 80
        assert np.allclose(p.proposal_boxes.tensor.numpy(), wb, atol=1e-4)
        assert np.array_equal(p.objectness_logits.numpy(), ws) and np.array_equal(p.oh_labels.numpy(), wo)
        assert p.oh_labels.dtype == torch.int32 and d["superpixels"].dtype == torch.int32
        sp = raw["superpixels"].repeat(2, 0).repeat(2, 1)
        assert np.array_equal(d["superpixels"].numpy(), sp[:, ::-1] if flip else sp)
    with pytest.raises(AssertionError):
        D.transform_proposals_seg({"proposal_file": path, "image_id": 3}, (240, 320), D.TransformList(tf),
                                  proposal_topk=50)


def test_xywh_proposal_files(tmp_path):
    path = os.path.join(str(tmp_path), "q.pkl")
    D.write_proposal_file(path, boxes=np.array([[10, 20, 30, 40]], np.float32), scores=np.ones((1,), np.float32),
                          oh_labels=np.ones((1, 4), np.uint8), superpixels=np.zeros((80, 60), np.int32), image_id="a",
                          bbox_mode=1)
    d = {"proposal_file": path, "image_id": "a"}
    D.transform_proposals_seg(d, (80, 60), D.TransformList([D.NoOpTransform()]), proposal_topk=10)
    assert d["proposals"].proposal_boxes.tensor.tolist() == [[10.0, 20.0, 40.0, 60.0]]


def test_dataset_mapper_contract(tmp_path):
    rng = np.random.RandomState(3)
    path, _ = _write_case(tmp_path, rng)
    img = rng.randint(0, 256, (120, 160, 3)).astype(np.uint8)
    sem = rng.randint(0, 54, (120, 160)).astype(np.uint8)
    mapper = D.DatasetMapper(True, augmentations=[D.ResizeShortestEdge((240, 240), 1333), D.RandomFlip(prob=1.0)],
                             precomputed_proposal_topk=60)
    out = mapper({"image_array": img, "sem_seg_array": sem, "proposal_file": path, "image_id": 17, "height": 120,
                  "width": 160, "annotations": [{"category_id": 3}, {"category_id": 9, "iscrowd": 1}, {"category_id": 3}]})
    assert out["image"].shape == (3, 240, 320) and out["image"].dtype == torch.uint8
    assert out["sem_seg"].shape == (240, 320) and out["sem_seg"].dtype == torch.int64
    assert out["superpixels"].shape == (240, 320) and len(out["proposals"]) <= 60
    assert out["instances"].gt_classes.tolist() == [3, 3]
    assert np.array_equal(out["sem_seg"].numpy(), sem.repeat(2, 0).repeat(2, 1)[:, ::-1])
    test = D.DatasetMapper(False, augmentations=[D.ResizeShortestEdge((120, 120), 1333)], precomputed_proposal_topk=60)
    o2 = test({"image_array": img, "proposal_file": path, "image_id": 17, "annotations": []})
    assert "instances" not in o2 and o2["image"].shape == (3, 120, 160)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_preprocess_images_u8_is_bit_exact(cuda):
    from jtsm_amd.layers.postprocess import preprocess_images_u8
    g = torch.Generator().manual_seed(0)
    imgs = [torch.randint(0, 256, (3, h, w), generator=g, dtype=torch.uint8) for h, w in ((37, 61), (64, 40), (1, 1))]
    mean, std = [102.9801, 115.9465, 122.7717], [1.0, 57.375, 58.395]
    out, sizes = preprocess_images_u8([i.cuda() for i in imgs], mean, std, size_divisibility=32)
    assert out.shape == (3, 3, 64, 64) and out.is_contiguous(memory_format=torch.channels_last)
    assert sizes == [(37, 61), (64, 40), (1, 1)]
    m, s = torch.tensor(mean).view(3, 1, 1), torch.tensor(std).view(3, 1, 1)
    for b, im in enumerate(imgs):
        want = torch.zeros(3, 64, 64)
        want[:, : im.shape[1], : im.shape[2]] = (im - m) / s          # mcnn.py:311 on the uint8 image
        assert torch.equal(out[b].cpu(), want)


@pytest.mark.gpu
def test_preprocess_images_f32_is_bit_exact(cuda):
    """Float32 images through the same launch: the bits of `(x - mean) / std` + ImageList.from_tensors."""
    from jtsm_amd.layers.postprocess import preprocess_images
    from jtsm_amd.structures import ImageList
    g = torch.Generator().manual_seed(1)
    imgs = [torch.rand(3, h, w, generator=g) * 255 for h, w in ((37, 61), (64, 40), (1, 1))]
    mean, std = [102.9801, 115.9465, 122.7717], [1.0, 57.375, 58.395]
    out, sizes = preprocess_images([i.cuda() for i in imgs], mean, std, size_divisibility=32)
    m, s = torch.tensor(mean).view(3, 1, 1), torch.tensor(std).view(3, 1, 1)
    want = ImageList.from_tensors([(im - m) / s for im in imgs], 32, channels_last=True)
    assert sizes == want.image_sizes and out.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(out.cpu(), want.tensor)
    with pytest.raises(TypeError):
        preprocess_images([imgs[0].cuda().double()], mean, std)


@pytest.mark.gpu
def test_prefetcher_delivers_identical_batches_and_model_trains(cuda):
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from model_util import jtsm_cfg, to_batched_inputs
    from jtsm_amd.modeling import build_model
    from oracle import model as OM
    batches = []
    for seed in (1, 2, 3):
        b = to_batched_inputs(OM.synthetic_batch(seed, B=2, size=128, R=40, sp_block=8))
        for x in b:
            x["image"] = x["image"].to(torch.uint8)               # what the mapper emits
            x["file_name"] = "img%d" % seed                       # non-tensor entries pass through
        batches.append(b)
    pre = D.DevicePrefetcher(batches, "cuda", depth=2)
    seen = 0
    model = build_model(jtsm_cfg("cuda"))
    model.train()
    model.roi_heads.box_head.dropout_p = 0.0
    for host, dev in zip(batches, pre):
        for h, d in zip(host, dev):
            assert d["file_name"] == h["file_name"] and d["image"].is_cuda and d["image"].dtype == torch.uint8
            assert torch.equal(d["image"].cpu(), h["image"]) and torch.equal(d["sem_seg"].cpu(), h["sem_seg"])
            assert d["sem_seg"].dtype == torch.int64 and d["proposals"].oh_labels.dtype == h["proposals"].oh_labels.dtype
            assert torch.equal(d["proposals"].oh_labels.cpu(), h["proposals"].oh_labels)
            assert torch.equal(d["proposals"].proposal_boxes.tensor.cpu(), h["proposals"].proposal_boxes.tensor)
            assert torch.equal(d["superpixels"].cpu(), h["superpixels"])
            assert torch.equal(d["instances"].gt_classes.cpu(), h["instances"].gt_classes)
        losses = model(dev)                                        # uint8 images -> fused preprocess launch
        assert all(torch.isfinite(v) for v in losses.values())
        seen += 1
    assert seen == 3 and pre.bytes_last > 0
    # same losses as the float path on the same (integer-valued) image
    ref = model([dict(x, image=x["image"].float()) for x in batches[2]])
    got = model(next(iter(D.DevicePrefetcher([batches[2]], "cuda"))))
    for k in ref:
        assert abs(float(ref[k].detach()) - float(got[k].detach())) <= 1e-6 * max(1.0, abs(float(ref[k].detach()))), k


@pytest.mark.gpu
def test_prefetcher_depth3_keeps_widened_tensors_intact_under_a_long_step(cuda):
    """at depth >= 3 the widening of batch i+2 runs while step i may still be reading ITS widened tensors;
    they were allocated on the side stream, so the allocator may only recycle them once the compute stream is done
    (record_stream on the CONSUMER stream).  A long-running 'step' (a big matmul chain queued on the compute stream
    before the labels are read) makes a premature reuse visible as corrupted labels."""
    g = torch.Generator().manual_seed(0)
    batches = []
    for k in range(6):
        sem = torch.randint(0, 54, (256, 256), generator=g)
        oh = torch.randint(0, 2, (300, 1024), generator=g, dtype=torch.int32)
        batches.append([{"image": torch.randint(0, 255, (3, 256, 256), generator=g, dtype=torch.uint8), "sem_seg": sem,
                         "oh_labels": oh}])
    pre = D.DevicePrefetcher(batches, "cuda", depth=3)
    a = torch.randn(4096, 4096, device=cuda)
    for host, dev in zip(batches, pre):
        x = a
        for _ in range(6):                       # ~ tens of ms of queued compute ahead of the reads below
            x = x @ a * 1e-3
        sem_d, oh_d = dev[0]["sem_seg"] + 0, dev[0]["oh_labels"] + 0      # read on the compute stream, after the matmuls
        assert sem_d.dtype == torch.int64
        assert torch.equal(sem_d.cpu(), host[0]["sem_seg"]) and torch.equal(oh_d.cpu(), host[0]["oh_labels"])
