"""CPU suite: the configs[0] (Faster R-CNN R50-FPN) pieces — oracle/rcnn.py and the product's device-agnostic label
logic — against the known answers the reference's own tests hold: the anchor tables of
tests/modeling/test_anchor_generator.py:17-75 and the matcher table of tests/modeling/test_matcher.py:19-30."""
import os

import pytest
import torch

from conftest import ROOT
from oracle import rcnn as OR

ANCHORS_OFFSET0 = torch.tensor([
    [-32.0, -8.0, 32.0, 8.0], [-16.0, -16.0, 16.0, 16.0], [-8.0, -32.0, 8.0, 32.0], [-64.0, -16.0, 64.0, 16.0],
    [-32.0, -32.0, 32.0, 32.0], [-16.0, -64.0, 16.0, 64.0], [-28.0, -8.0, 36.0, 8.0], [-12.0, -16.0, 20.0, 16.0],
    [-4.0, -32.0, 12.0, 32.0], [-60.0, -16.0, 68.0, 16.0], [-28.0, -32.0, 36.0, 32.0], [-12.0, -64.0, 20.0, 64.0]])
ANCHORS_OFFSET_HALF = ANCHORS_OFFSET0 + 2.0     # test_anchor_generator.py:51-75: every coordinate + stride / 2


def test_oracle_anchors_reproduce_reference_tables():
    cell = torch.cat([OR.cell_anchors(s, (0.25, 1, 4)) for s in (32, 64)])
    assert torch.allclose(OR.grid_anchors(1, 2, 4, cell, 0.0), ANCHORS_OFFSET0)
    assert torch.allclose(OR.grid_anchors(1, 2, 4, cell, 0.5), ANCHORS_OFFSET_HALF)


def test_product_anchor_generator_reproduces_reference_tables():
    from jtsm_amd.modeling.anchor_generator import DefaultAnchorGenerator

    feat = [torch.rand(2, 96, 1, 2)]
    gen = DefaultAnchorGenerator(sizes=[[32, 64]], aspect_ratios=[[0.25, 1, 4]], strides=[4], offset=0.0)
    assert torch.allclose(gen(feat)[0].tensor, ANCHORS_OFFSET0) and gen.num_anchors == [6]
    gen = DefaultAnchorGenerator(sizes=[32, 64], aspect_ratios=[0.25, 1, 4], strides=[4])     # explicit args, offset 0.5
    assert torch.allclose(gen(feat)[0].tensor, ANCHORS_OFFSET_HALF)


def test_matchers_reproduce_reference_table():
    from jtsm_amd.modeling.matcher import Matcher

    q = torch.tensor([[0.15, 0.45, 0.2, 0.6], [0.3, 0.65, 0.05, 0.1], [0.05, 0.4, 0.25, 0.4]])
    want_idx, want_lab = torch.tensor([1, 1, 2, 0]), torch.tensor([-1, 1, 0, 1], dtype=torch.int8)
    idx, lab = OR.match(q, (0.3, 0.7), (0, -1, 1), True)
    assert torch.equal(idx, want_idx) and torch.equal(lab, want_lab)
    idx, lab = Matcher([0.3, 0.7], [0, -1, 1], allow_low_quality_matches=True)(q)
    assert torch.equal(idx, want_idx) and torch.equal(lab, want_lab)


def test_oracle_faster_rcnn_step_is_finite_and_deterministic():
    p = OR.init_params(seed=2, input_gain=1.0 / 64, head_gain=0.02)
    b = OR.synthetic_batch(7)
    l0 = OR.forward_losses(p, b, rpn_batch=100000, roi_batch=100000)
    l1 = OR.forward_losses(p, b, rpn_batch=100000, roi_batch=100000)
    assert set(l0) == {"loss_rpn_cls", "loss_rpn_loc", "loss_cls", "loss_box_reg"}
    assert all(bool(torch.isfinite(v)) and float(v) == float(l1[k]) for k, v in l0.items())
    with pytest.raises(AssertionError):            # sub-sampling would be random: the oracle refuses
        OR.forward_losses(p, b, rpn_batch=8, roi_batch=100000)


REF = "/root/reference/configs"


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists in the build container only")
@pytest.mark.parametrize("path,arch,mparams", [
    ("COCO-Detection/faster_rcnn_R_50_FPN_1x.yaml", "GeneralizedRCNN", 41.7),
    ("COCO-InstanceSegmentation/mask_rcnn_R_50_FPN_1x.yaml", "GeneralizedRCNN", 44.3),
    ("COCO-PanopticSegmentation/panoptic_fpn_R_50_1x.yaml", "PanopticFPN", 46.0)])
def test_reference_detection_configs_build_unchanged(path, arch, mparams):
    """BASELINE configs[0] and its two siblings load through their _BASE_ chain and build through the registries with
    the published parameter counts (MODEL_ZOO: 41.7 M / 44.4 M / 46.0 M)."""
    from jtsm_amd.config import get_cfg
    from jtsm_amd.modeling import build_model

    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(REF, path))
    cfg.MODEL.DEVICE = "cpu"
    model = build_model(cfg)
    assert type(model).__name__ == arch
    assert abs(sum(p.numel() for p in model.parameters()) / 1e6 - mparams) < 0.15
    flat = get_cfg()
    flat.merge_from_file(os.path.join(ROOT, "configs", "faster_rcnn_R_50_FPN_1x.yaml"))
    if path.startswith("COCO-Detection"):
        for sect in ("RPN", "ROI_HEADS", "ROI_BOX_HEAD", "ANCHOR_GENERATOR", "FPN", "RESNETS"):
            assert dict(flat.MODEL[sect]) == dict(cfg.MODEL[sect]), sect
