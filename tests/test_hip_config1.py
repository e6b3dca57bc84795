"""GPU suite: BASELINE configs[1] at its own size — R50-FPN forward only on 8 x 3 x 1024 x 1024 random tensors, then
ROIAlign on 512 synthetic proposals over the pyramid it produced.  Too big for the torch-CPU oracle's backbone, so the
backbone is checked through properties that need none, and the pooling — whose C oracle finishes in seconds at this
size — against oracle/pooling.py on the very feature maps the HIP backbone wrote."""
import math

import numpy as np
import pytest
import torch

from model_util import jtsm_cfg
from oracle import pooling as OP

pytestmark = pytest.mark.gpu

from jtsm_amd.layers import conv as K  # noqa: E402
from jtsm_amd.modeling import build_model  # noqa: E402
from jtsm_amd.modeling.poolers import ROIPooler, assign_boxes_to_levels  # noqa: E402
from jtsm_amd.structures import Boxes  # noqa: E402

LEVELS = ("p2", "p3", "p4", "p5")


def _backbone(cuda):
    torch.manual_seed(0)
    model = build_model(jtsm_cfg("cuda"))
    model.eval()
    with torch.no_grad():
        model.backbone.bottom_up.stem.conv1.weight.mul_(1.0 / 64)
    return model


def _forward(model, images):
    K.planes_clear()
    with torch.no_grad():
        return {k: v.float() for k, v in model.backbone(images).items()}


def test_config1_forward_only_full_size(cuda):
    B, S = 8, 1024
    model = _backbone(cuda)
    g = torch.Generator().manual_seed(1234)
    images = ((torch.rand(B, 3, S, S, generator=g) * 255).to(cuda) - model.pixel_mean).contiguous(
        memory_format=torch.channels_last)
    old = K.MATH
    try:
        K.set_math("bf16x3")
        feats = _forward(model, images)
        # shapes of the pyramid (fpn.py:114-152): 256 channels at strides 4 .. 64
        for name, stride in zip(LEVELS + ("p6",), (4, 8, 16, 32, 64)):
            assert tuple(feats[name].shape) == (B, 256, S // stride, S // stride), name
            assert bool(torch.isfinite(feats[name]).all()) and float(feats[name].abs().max()) > 0
        # (1) images of a batch do not see each other: image 5 alone gives the rows it gave inside the batch (tile
        # and split-K plans depend on the batch size, i.e. the summation order of all 53 layers differs: measured
        # 2.2e-5 of the map's magnitude; bar = the north-star 1e-4)
        alone = _forward(model, images[5:6])
        for name in LEVELS:
            a, b = alone[name][0], feats[name][5]
            assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()), name
        # (2) a second run reproduces the batch bit for bit (deterministic kernels, no atomics on this path)
        again = _forward(model, images)
        for name in LEVELS:
            assert torch.equal(again[name], feats[name]), name
        # (3) the two arithmetics agree at the north-star bar: split-bf16 against exact fp32 MFMA, 1e-4 of each map's
        # magnitude through all 53 convolutions + FPN
        K.set_math("f32")
        exact = _forward(model, images[:2])
        for name in LEVELS:
            a, b = feats[name][:2], exact[name]
            assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()), (name, float((a - b).abs().max()),
                                                                                float(b.abs().max()))
    finally:
        K.set_math(old)

    # ---- ROIAlign on 512 synthetic proposals (64 per image, SURVEY §8d recipe) over that pyramid, forward + backward
    boxes = []
    for _ in range(B):
        n = 64
        x0, y0 = torch.rand(n, generator=g) * S * 0.75, torch.rand(n, generator=g) * S * 0.75
        lo, hi = math.log(16.0), math.log(S / 2.0)
        w = torch.exp(torch.rand(n, generator=g) * (hi - lo) + lo)
        h = torch.exp(torch.rand(n, generator=g) * (hi - lo) + lo)
        boxes.append(Boxes(torch.stack([x0, y0, (x0 + w).clamp(max=S), (y0 + h).clamp(max=S)], 1).to(cuda)))
    levels = [feats[n].contiguous(memory_format=torch.channels_last) for n in LEVELS]
    lv = assign_boxes_to_levels(boxes, 2, 5, 224, 4).cpu().numpy()
    assert len(np.unique(lv)) >= 3                          # the proposals span the pyramid
    rois = torch.cat([torch.cat([torch.full((len(b), 1), float(i), device=cuda), b.tensor], 1)
                      for i, b in enumerate(boxes)]).cpu().numpy()
    for res in (7, 14):
        pooler = ROIPooler(res, [1 / 4, 1 / 8, 1 / 16, 1 / 32], 0, "ROIAlignV2")
        xs = [f.clone().requires_grad_() for f in levels]
        y = pooler(xs, boxes)
        assert tuple(y.shape) == (512, 256, res, res)
        got = y.detach().cpu().numpy()
        for l, f in enumerate(levels):                      # bit-exact against the C oracle, level by level
            sel = np.nonzero(lv == l)[0]
            if sel.size == 0:
                continue
            want = OP.roi_align_forward(np.ascontiguousarray(f.cpu().numpy()), rois[sel], 1.0 / (4 << l), res, res,
                                        0, True)
            assert np.array_equal(got[sel], want), (res, l)
        # backward of ones: every bin hands out weight 1 (aligned sampling, border samples clamp inside), so the
        # gradient maps sum to rois x channels x bins
        y.backward(torch.ones_like(y))
        total = sum(float(x.grad.double().sum()) for x in xs)
        assert abs(total - 512 * 256 * res * res) <= 1e-3 * 512 * 256 * res * res
