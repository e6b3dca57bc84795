"""Build-container only (skipped where /root/reference is absent): every JTSM configuration the reference ships
(projects/WSL/configs/*PanopticSegmentation/jtsm_*.yaml) loads UNCHANGED through its _BASE_ chain and builds
through the registries; the flattened copies under configs/ (what the GPU tests use — the reference tree does not
travel) say the same as the reference-merged ones on every MODEL / WSL key."""
import glob
import os

import pytest

from conftest import ROOT

REF = "/root/reference/projects/WSL/configs"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists in the build container only")


def _cfg(path):
    from jtsm_amd.config import add_wsl_config, get_cfg
    cfg = get_cfg()
    add_wsl_config(cfg)
    cfg.merge_from_file(path)
    cfg.MODEL.DEVICE = "cpu"
    return cfg


def _flat(node, pre=""):
    out = {}
    for k, v in node.items():
        if hasattr(v, "items"):
            out.update(_flat(v, pre + k + "."))
        else:
            out[pre + k] = v
    return out


SHIPPED = sorted(glob.glob(os.path.join(REF, "*PanopticSegmentation", "jtsm_*.yaml")))


def test_there_are_four_shipped_jtsm_configs():
    assert [os.path.basename(p) for p in SHIPPED] == [
        "jtsm_WSR_18_DC5_1x.yaml", "jtsm_WSR_18_DC5_1x.yaml", "jtsm_WSR_18_DC5_1x_VOC2007.yaml", "jtsm_WSR_50_DC5_1x.yaml"]


@pytest.mark.parametrize("path", SHIPPED, ids=["/".join(p.split("/")[-2:]) for p in SHIPPED])
def test_shipped_config_builds_unchanged(path):
    from jtsm_amd.modeling import build_model
    from jtsm_amd.modeling.backbone.resnet_wsl_v2 import PooledBasicBlock, PooledBottleneckBlock

    cfg = _cfg(path)
    model = build_model(cfg)
    depth = cfg.MODEL.RESNETS.DEPTH
    want = PooledBasicBlock if depth in (18, 34) else PooledBottleneckBlock
    blocks = [b for stage in model.backbone.stages for b in stage]
    assert blocks and all(type(b) is want for b in blocks)
    assert model.backbone.output_shape()["res5"].stride == 8                    # dilated C5
    assert model.backbone.output_shape()["res5"].channels == (512 if depth in (18, 34) else 2048)
    assert not any(p.requires_grad for p in model.backbone.parameters())         # FREEZE_AT 5
    v1 = cfg.MODEL.BACKBONE.NAME == "build_wsl_resnet_backbone"
    pooled = [(n, i) for n, stage in zip(model.backbone.stage_names, model.backbone.stages)
              for i, b in enumerate(stage) if b.has_pool]
    if v1:   # pools behind the LAST block of res2 / res3 (resnet_wsl.py:684-696)
        assert pooled == [("res2", len(model.backbone.res2) - 1), ("res3", len(model.backbone.res3) - 1)]
        assert all(b.pool_output for stage in model.backbone.stages for b in stage if b.has_pool)
    else:    # in front of the FIRST block of res3 / res4 (resnet_wsl_v2.py:694-701)
        assert pooled == [("res3", 0), ("res4", 0)]
    assert (model.roi_heads.mask_on) == bool(cfg.MODEL.MASK_ON)
    assert model.roi_heads.refine_K == 4 and model.roi_heads.num_classes == cfg.MODEL.ROI_HEADS.NUM_CLASSES


@pytest.mark.parametrize("depth", [18, 50])
def test_flattened_config_equals_reference_chain(depth):
    mine = _flat(_cfg(os.path.join(ROOT, "configs", "jtsm_WSR_%d_DC5_1x.yaml" % depth)))
    ref = _flat(_cfg(os.path.join(REF, "PascalVOC-PanopticSegmentation", "jtsm_WSR_%d_DC5_1x.yaml" % depth)))
    skip = ("MODEL.WEIGHTS", "MODEL.DEVICE")     # checkpoint path: outside the hot path
    diff = {k: (mine.get(k), ref.get(k)) for k in set(mine) | set(ref)
            if k.startswith(("MODEL.", "WSL.")) and k not in skip and k in mine and mine.get(k) != ref.get(k)}
    assert not diff, diff
