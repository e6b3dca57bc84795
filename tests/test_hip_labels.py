"""Label-preparation kernels (csrc/mining.hip, round 3) against the tensor-op definitions they replace in the step:
ROIPooler's box format + level assignment (detectron2/modeling/poolers.py:22-95), the box head's per-roi factor
(roi_heads_jtsm.py:607-633) and the image-level labels (roi_heads.py:146-161).  Bit-exact: integer outputs and float
expressions evaluated operation by operation."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cuda():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    return torch.device("cuda", 0)


def _boxes(g, n, size=1024.0):
    xy = torch.rand(n, 2, generator=g) * size * 0.8
    wh = torch.rand(n, 2, generator=g) ** 2 * size * 0.6 + 1.0
    return torch.cat([xy, xy + wh], dim=1)


def test_pooler_rois_and_levels_are_the_tensor_op_ones(cuda):
    from jtsm_amd.layers.mining import pooler_rois_levels
    from jtsm_amd.modeling.poolers import assign_boxes_to_levels, convert_boxes_to_pooler_format
    from jtsm_amd.structures import Boxes

    g = torch.Generator().manual_seed(3)
    # sizes sitting on the level thresholds (sqrt(area) = 224 * 2^k / 2 ...), degenerate, negative and NaN boxes
    edge = []
    for s in (56.0, 111.99999, 112.0, 112.00001, 224.0, 447.99997, 448.0, 896.0, 0.0, 1e-6):
        edge.append([10.0, 20.0, 10.0 + s, 20.0 + s])
    edge += [[5.0, 5.0, 2.0, 9.0], [0.0, 0.0, float("nan"), 4.0], [0.0, 0.0, float("inf"), 4.0], [3.0, 3.0, 3.0, 3.0]]
    lists = [torch.cat([_boxes(g, 2000), torch.tensor(edge)]), torch.zeros(0, 4), _boxes(g, 777), _boxes(g, 1)]
    for (lo, hi) in ((2, 5), (3, 3), (2, 6)):
        dev = [Boxes(b.to(cuda)) for b in lists]
        rois, level = pooler_rois_levels([b.tensor for b in dev], lo, hi, 224, 4)
        want_rois = convert_boxes_to_pooler_format(dev)
        want_level = assign_boxes_to_levels(dev, lo, hi, 224, 4)
        assert torch.equal(rois.view(torch.int32), want_rois.view(torch.int32))       # (bit patterns: NaN included)
        assert torch.equal(level.to(torch.int64), want_level), (lo, hi, (level.to(torch.int64) != want_level).nonzero()[:5])
        assert int(level.min()) >= 0 and int(level.max()) <= hi - lo


def test_roi_scale_is_the_tensor_op_expression(cuda):
    from jtsm_amd.layers.mining import roi_scale

    g = torch.Generator().manual_seed(4)
    for (counts, ch, p) in (([2000, 2000], 256, 7), ([5, 0, 300], 8, 3), ([1], 4, 1)):
        m = sum(counts)
        arg = torch.randint(-1, 50, (m, ch, p, p), generator=g, dtype=torch.int32)
        arg[::3, 0] = -1
        arg[1::7, 0, : p // 2 + 1] = -1
        arg = arg.to(cuda).contiguous(memory_format=torch.channels_last)
        obj = [torch.randn(n, generator=g).to(cuda) for n in counts]
        got = roi_scale(arg, obj)
        bins = p * p
        nvalid = (arg[:, 0, :, :] != -1).reshape(m, -1).sum(dim=1).to(dtype=torch.float32)
        want = bins * (nvalid + 1).reciprocal()
        want = want * torch.cat([o + 1 for o in obj], dim=0)
        assert torch.equal(got, want), (got - want).abs().max().item()


@pytest.mark.parametrize("sem_dtype", [torch.int64, torch.uint8, None])
def test_image_labels_are_the_tensor_op_ones(cuda, sem_dtype):
    from jtsm_amd.layers.mining import image_labels
    from jtsm_amd.modeling.roi_heads.roi_heads_jtsm import class_lists, present_stuff, present_things

    class T:      # (what present_things reads of an Instances)
        def __init__(self, c):
            self.gt_classes = c

    g = torch.Generator().manual_seed(5)
    nc, ns = 80, 54
    gts = [torch.randint(0, nc, (n,), generator=g).to(cuda) for n in (7, 0, 40, 300)]
    sem = None
    if sem_dtype is not None:
        sem = torch.randint(0, ns, (len(gts), 96, 160), generator=g)
        sem[0] = 0                                   # an image of things only
        sem[1, :10] = 255                            # ignore
        sem[2][sem[2] > 5] = 7                       # few labels
        sem = sem.to(sem_dtype).to(cuda)
    oh_t, cls_t, cnt_t, oh_s, cls_s, cnt_s = image_labels(gts, nc, sem, ns, nc)
    want_oh = present_things([T(c) for c in gts], nc)
    want_cls, want_cnt = class_lists(want_oh)
    assert torch.equal(oh_t, want_oh) and torch.equal(cls_t, want_cls) and torch.equal(cnt_t, want_cnt)
    assert cls_t.is_contiguous() and cls_t.dtype == torch.int32 and cnt_t.dtype == torch.int32
    if sem is None:
        assert oh_s is None and cls_s is None and cnt_s is None
        return
    want_oh_s = present_stuff(sem.to(torch.int64), ns)
    want_cls_s, want_cnt_s = class_lists(want_oh_s, offset=nc)
    assert torch.equal(oh_s, want_oh_s) and torch.equal(cls_s, want_cls_s) and torch.equal(cnt_s, want_cnt_s)
    assert int(cnt_s[0]) == 0


def test_label_preparation_rejects_too_many_images(cuda):
    from jtsm_amd.layers.mining import MAX_IMAGES, pooler_rois_levels

    with pytest.raises(RuntimeError):
        pooler_rois_levels([torch.zeros(1, 4, device=cuda)] * (MAX_IMAGES + 1), 2, 5, 224, 4)


def test_fg_compact_is_nonzero_plus_gathers(cuda):
    from jtsm_amd.layers.mining import fg_compact

    g = torch.Generator().manual_seed(6)
    for (counts, p_fg) in (([2000, 2000], 0.08), ([5, 0, 1300, 1], 0.5), ([3000], 0.0), ([1100], 1.0)):
        R, bg = sum(counts), 80
        labels = torch.where(torch.rand(R, generator=g) < p_fg, torch.randint(0, bg, (R,), generator=g), torch.tensor(bg))
        labels = labels.to(torch.int32).to(cuda)
        boxes = _boxes(g, R).to(cuda)
        matched = torch.randint(0, 100, (R,), generator=g, dtype=torch.int32).to(cuda)
        offs = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32).to(cuda)
        out = fg_compact(labels, bg, offs, boxes, matched)
        fg = torch.nonzero(labels != bg)[:, 0]
        n = fg.numel()
        per_image = [int(((fg >= offs[b]) & (fg < offs[b + 1])).sum()) for b in range(len(counts))]
        assert out["counts"].tolist() == per_image and sum(per_image) == n
        img = torch.bucketize(fg, offs[1:].to(torch.int64), right=True)
        assert torch.equal(out["rows"][:n].to(torch.int64), fg)
        assert torch.equal(out["boxes"][:n], boxes[fg]) and torch.equal(out["classes"][:n], labels[fg].to(torch.int64))
        assert torch.equal(out["img"][:n].to(torch.int64), img) and torch.equal(out["matched"][:n], matched[fg])
        assert torch.equal(out["rois"][:n], torch.cat([img.to(torch.float32)[:, None], boxes[fg]], dim=1))
        assert fg_compact(labels, bg, offs, boxes)["matched"] is None
