"""GPU suite: test-time augmentation with averaging (projects/WSL/wsl/modeling/test_time_augmentation_avg.py) on the
HIP path.  The per-view passes are the model's own inference (covered by test_hip_inference.py); checked here are the
mapper, the device-side inverse transforms / reductions against the reference's literal host expressions
(`tfm.inverse().apply_box(numpy)`, `F.interpolate(nearest)` + `apply_segmentation`, flip + mean), and that TTA over
the single identity view reproduces plain inference exactly."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from model_util import jtsm_cfg, to_batched_inputs
from oracle import model as OM

pytestmark = pytest.mark.gpu

from jtsm_amd.data.transforms import HFlipTransform  # noqa: E402
from jtsm_amd.modeling import build_model  # noqa: E402
from jtsm_amd.modeling.test_time_augmentation_avg import (DatasetMapperTTAAVG, GeneralizedRCNNWithTTAAVG,  # noqa: E402
                                                          apply_box_device)


@pytest.fixture(scope="module")
def setup(cuda):
    torch.manual_seed(0)
    params = OM.init_params(seed=3, random_bn=True, input_gain=1.0 / 64)
    batch = OM.synthetic_batch(77, B=1, size=128, R=60, sp_block=8)
    cfg = jtsm_cfg("cuda")
    cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST = 1e-5
    cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST = 0.3
    cfg.MODEL.PANOPTIC_FPN.COMBINE.INSTANCES_CONFIDENCE_THRESH = 0.02
    cfg.MODEL.PANOPTIC_FPN.COMBINE.STUFF_AREA_LIMIT = 64
    cfg.TEST.AUG.MIN_SIZES = (128, 160)
    cfg.TEST.AUG.MAX_SIZE = 400
    cfg.TEST.AUG.FLIP = True
    cfg.DATASETS.PRECOMPUTED_PROPOSAL_TOPK_TEST = 1000
    model = build_model(cfg)
    model.load_state_dict({k: v.detach() for k, v in params.items()}, strict=True)
    model.eval()
    inp = to_batched_inputs(batch)[0]
    inp["image"] = inp["image"].clamp(0, 255).to(torch.uint8)          # what the dataset mapper emits
    inp["height"], inp["width"] = 128, 128
    return cfg, model, inp


def test_mapper_views(setup):
    cfg, _, inp = setup
    views = DatasetMapperTTAAVG(cfg)(inp)
    assert len(views) == 4                                             # 2 sizes x (plain, mirrored)
    assert [tuple(v["image"].shape) for v in views] == [(3, 128, 128), (3, 128, 128), (3, 160, 160), (3, 160, 160)]
    plain, mirrored = views[2], views[3]
    assert torch.equal(mirrored["image"], plain["image"].flip(2))
    assert torch.equal(mirrored["superpixels"], plain["superpixels"].flip(1))
    pb, mb = plain["proposals"].proposal_boxes.tensor, mirrored["proposals"].proposal_boxes.tensor
    assert torch.allclose(mb[:, 0], 160 - pb[:, 2], atol=1e-4) and torch.allclose(mb[:, 2], 160 - pb[:, 0], atol=1e-4)
    assert torch.allclose(pb, inp["proposals"].proposal_boxes.tensor * 1.25, atol=1e-4)
    assert torch.equal(plain["proposals"].oh_labels, inp["proposals"].oh_labels)
    assert any(isinstance(t, HFlipTransform) for t in mirrored["transforms"].transforms)


def test_device_box_transform_matches_host_apply_box(setup):
    cfg, _, inp = setup
    views = DatasetMapperTTAAVG(cfg)(inp)
    g = torch.Generator().manual_seed(1)
    boxes = torch.rand(50, 4, generator=g) * 60
    boxes[:, 2:] += boxes[:, :2]
    for v in views:
        for tfm in (v["transforms"], v["transforms"].inverse()):
            want = tfm.apply_box(boxes.numpy().copy())
            got = apply_box_device(tfm, boxes.cuda()).cpu().numpy()
            assert np.allclose(got, want, rtol=1e-6, atol=1e-4)


def test_identity_view_reproduces_plain_inference(setup):
    cfg, model, inp = setup
    one = cfg.clone()
    one.TEST.AUG.MIN_SIZES = (128,)
    one.TEST.AUG.FLIP = False
    tta = GeneralizedRCNNWithTTAAVG(one, model)
    got = tta([inp])[0]
    want = model.inference([inp])[0]
    assert torch.equal(got["instances"].pred_boxes.tensor, want["instances"].pred_boxes.tensor)
    assert torch.equal(got["instances"].scores, want["instances"].scores)
    assert torch.equal(got["instances"].pred_classes, want["instances"].pred_classes)
    assert torch.equal(got["instances"].pred_masks, want["instances"].pred_masks)
    assert torch.equal(got["sem_seg"], want["sem_seg"])
    assert torch.equal(got["panoptic_seg"][0], want["panoptic_seg"][0]) and got["panoptic_seg"][1] == want["panoptic_seg"][1]


def test_reductions_match_the_reference_host_expressions(setup):
    cfg, model, inp = setup
    tta = GeneralizedRCNNWithTTAAVG(cfg, model)
    views = tta.tta_mapper(inp)
    tfms = [v.pop("transforms") for v in views]
    # ---- boxes / scores (test_time_augmentation_avg.py:365-392)
    with tta._turn_off_roi_heads(["mask_on"]):
        _, all_scores, all_boxes = tta._batch_inference(views)
        boxes, scores = tta._get_augmented_boxes(views, tfms)
    assert model.roi_heads.mask_on
    back = []
    for b, t in zip(all_boxes, tfms):
        n, r, c = b.shape
        back.append(torch.from_numpy(t.inverse().apply_box(b.reshape(r * c // 4, 4).cpu().numpy())).reshape(1, r, c))
    want_boxes = torch.mean(torch.cat(back, 0), 0)
    assert torch.allclose(boxes.cpu(), want_boxes.float(), rtol=1e-5, atol=1e-3)
    assert torch.allclose(scores.cpu(), torch.mean(torch.cat(all_scores, 0), 0).cpu(), rtol=1e-6, atol=1e-9)
    # ---- masks (:418-426)
    merged = tta._merge_detections(boxes, scores, (128, 128))
    assert len(merged) > 0
    outs, _, _ = tta._batch_inference(views, tta._rescale_detected_boxes(views, merged, tfms))
    red = tta._reduce_pred_masks(outs, tfms)
    lit = []
    for o, t in zip(outs, tfms):
        m = o.pred_masks.cpu()
        lit.append(m.flip(dims=[3]) if any(isinstance(x, HFlipTransform) for x in t.transforms) else m)
    assert torch.allclose(red.cpu(), torch.mean(torch.stack(lit, 0), 0), rtol=1e-6, atol=1e-7)
    # ---- semantic logits (:428-442)
    souts, _, _ = tta._batch_inference(views, only_sem_seg=True)
    sem = tta._reduce_pred_sem_seg(views, souts, tfms, (128, 128))
    lit = []
    for v, o, t in zip(views, souts, tfms):
        h, w = v["image"].shape[1:3]
        x = F.interpolate(o.cpu().unsqueeze(0), size=(h, w), mode="nearest").squeeze(0)
        s = t.inverse().apply_segmentation(x.permute(1, 2, 0).numpy())
        lit.append(torch.from_numpy(np.ascontiguousarray(s)).permute(2, 0, 1))
    want = torch.mean(torch.stack(lit, 0), 0)
    assert sem.shape == (54, 128, 128) and torch.allclose(sem.cpu(), want, rtol=1e-6, atol=1e-6)


def test_full_tta_output_contract(setup):
    cfg, model, inp = setup
    out = GeneralizedRCNNWithTTAAVG(cfg, model)([inp])[0]
    assert set(out) == {"instances", "sem_seg", "panoptic_seg"}
    assert out["instances"].pred_masks.shape[1:] == (128, 128) and out["sem_seg"].shape == (54, 128, 128)
    assert out["panoptic_seg"][0].shape == (128, 128) and len(out["panoptic_seg"][1]) > 0
