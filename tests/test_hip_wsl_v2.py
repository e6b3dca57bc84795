"""GPU suite: the shipped JTSM configuration — ResNet-WS v2, dilated C5, frozen backbone,
single-level MOIPool — on the HIP path.  The backbone is checked against the same architecture written with
stock torch operators on the CPU (fp64) from the same weights; the composite is run for one training step."""
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import ROOT

pytestmark = pytest.mark.gpu

from jtsm_amd.config import add_wsl_config, get_cfg  # noqa: E402
from jtsm_amd.layers.elementwise import max_pool_2x2  # noqa: E402
from jtsm_amd.modeling import build_model  # noqa: E402
from jtsm_amd.utils.synthetic import synthetic_inputs  # noqa: E402


def dc5_cfg(device, depth=50):
    cfg = get_cfg()
    add_wsl_config(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "jtsm_WSR_%d_DC5_1x.yaml" % depth))
    cfg.MODEL.DEVICE = device
    return cfg


@pytest.mark.parametrize("stride", [1, 2])
def test_max_pool_2x2_matches_torch(cuda, stride):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 8, 9, 11, generator=g)
    x[0, :, 2:4, 2:4] = 1.5            # ties inside windows: the first maximum must take the gradient
    x0 = x.clone().double().requires_grad_(True)
    y0 = F.max_pool2d(x0, 2, 2) if stride == 2 else F.max_pool2d(F.pad(x0, (0, 1, 0, 1)), 2, 1)
    gy = torch.randn(y0.shape, generator=g)
    y0.backward(gy.double())
    xd = x.to(cuda).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = max_pool_2x2(xd, stride)
    assert torch.equal(y.cpu(), y0.detach().float())
    y.backward(gy.to(cuda))
    assert torch.allclose(xd.grad.cpu(), x0.grad.float(), rtol=0, atol=1e-6)


def _bn(x, norm):
    scale = norm.weight * (norm.running_var + norm.eps).rsqrt()
    return x * scale.view(1, -1, 1, 1) + (norm.bias - norm.running_mean * scale).view(1, -1, 1, 1)


def _conv(x, m):
    return _bn(F.conv2d(x, m.weight, None, m.stride, m.padding, m.dilation), m.norm)


def stock_v2_forward(backbone, x):
    """The ResNet-WS v2 forward written with stock operators (resnet_wsl_v2.py:230-251,418-429,499-523)."""
    s = backbone.stem
    x = F.relu(_conv(x, s.conv1)); x = F.relu(_conv(x, s.conv2)); x = F.relu(_conv(x, s.conv3))
    x = F.max_pool2d(x, 2, 2)
    for stage in backbone.stages:
        for b in stage:
            if b.has_pool:
                x = F.max_pool2d(x, 2, 2) if b.pool_stride == 2 else F.max_pool2d(F.pad(x, (0, 1, 0, 1)), 2, 1)
            out = F.relu(_conv(x, b.conv1))
            if hasattr(b, "conv3"):
                out = F.relu(_conv(out, b.conv2)); out = _conv(out, b.conv3)
            else:
                out = _conv(out, b.conv2)
            x = F.relu(out + (_conv(x, b.shortcut) if b.shortcut is not None else x))
    return x


@pytest.mark.parametrize("depth", [50, 18])
def test_dc5_backbone_features_match_stock_torch(cuda, depth):
    torch.manual_seed(0)
    model = build_model(dc5_cfg("cuda", depth))
    bb = model.backbone
    with torch.no_grad():
        for m in bb.modules():           # non-trivial frozen-BN statistics
            if hasattr(m, "running_var"):
                m.running_var.uniform_(0.5, 1.5); m.running_mean.normal_(0, 0.1)
                m.weight.uniform_(0.8, 1.2); m.bias.normal_(0, 0.1)
        bb.stem.conv1.weight.mul_(1.0 / 64)
    c5 = 2048 if depth == 50 else 512
    assert bb.output_shape()["res5"].stride == 8 and bb.output_shape()["res5"].channels == c5
    assert not any(p.requires_grad for p in bb.parameters())           # FREEZE_AT 5
    x = torch.rand(2, 3, 96, 128) * 255
    y = bb(x.to(cuda).contiguous(memory_format=torch.channels_last))["res5"]
    assert tuple(y.shape) == (2, c5, 12, 16)
    ref = stock_v2_forward(bb.double().cpu(), x.double())
    err = (y.detach().cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    assert err <= 1e-4, err


def test_dc5_composite_training_step(cuda):
    """One training step of the shipped configuration: only head parameters train, every loss is finite, the
    (parameter-free) semantic head contributes none."""
    torch.manual_seed(0)
    model = build_model(dc5_cfg("cuda"))
    model.train()
    with torch.no_grad():
        model.backbone.stem.conv1.weight.mul_(1.0 / 64)
    inputs = synthetic_inputs(99, batch=2, size=256, proposals=120, sp_block=8, device=cuda, num_things=20, num_stuff=2,
                              n_stuff=1)
    losses = model(inputs)
    assert "loss_sem_seg" not in losses and {"loss_cls", "loss_mask", "loss_cls_r3", "loss_box_reg_r3"} <= set(losses)
    total = sum(losses.values())
    assert bool(torch.isfinite(total))
    total.backward()
    for n, p in model.named_parameters():
        if n.startswith("backbone."):
            assert p.grad is None, n
        elif p.requires_grad:
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()), n


@pytest.fixture
def conv_math(request):
    from jtsm_amd.layers import conv as K
    old = K.MATH
    K.set_math(request.param)
    yield request.param
    K.set_math(old)


# loss bar per (depth, convolution arithmetic).  The exact-fp32 MFMA path is the parity arithmetic: measured worst
# loss deviation 6.9e-6 at depth 50.  The split-bf16 path (16-17 mantissa bits per operand, the default of the bench)
# is inside 1e-4 at depth 18 but measured 1.30e-4 on ONE loss (loss_box_reg_r2, itself 1.1e-2; the others <= 8.6e-5)
# behind the 53 frozen convolutions of depth 50 — its bar there is 2e-4 and says so.
DC5_CASES = [(18, "bf16x3", 1e-4), (50, "f32", 1e-4), (50, "bf16x3", 2e-4)]


@pytest.mark.parametrize("depth,conv_math,loss_bar", DC5_CASES, indirect=["conv_math"],
                         ids=["%d-%s" % (d, m) for d, m, _ in DC5_CASES])
def test_dc5_composite_matches_oracle(cuda, depth, conv_math, loss_bar):
    """The shipped single-level configuration (ResNet-WS v2 18 = BasicBlock / 50 = Bottleneck, dilated C5, frozen
    backbone, single-level MOIPool + ROIAlign on res5, DAN, MIL + 4 refinements, two mask heads, TwoClassHead)
    against oracle/model.py's `dc5` architecture on the same seeded weights and batch: every loss at 1e-4 (see
    DC5_CASES), the integer artefacts (mined rows, labels, foreground set, pseudo semantic target) bit-exact, head
    gradients."""
    from model_util import to_batched_inputs
    from oracle import model as OM

    dan = (4096, 4096) if depth == 18 else (2048, 4096)
    params = OM.init_params_dc5(seed=11, depth=depth, nt=20, ns=2, dan_dims=dan, input_gain=1.0 / 64)
    batch = OM.synthetic_batch(77, B=2, size=256, R=120, sp_block=8, n_stuff=1, nt=20, ns=2)
    names = [k for k in params if k.startswith("roi_heads.")]
    for n in names:
        params[n].requires_grad_(True)
    losses0, aux0 = OM.forward_losses(params, batch, depth=depth, return_aux=True, arch="dc5", nt=20, ns=2)
    sum(losses0.values()).backward()

    model = build_model(dc5_cfg("cuda", depth))
    missing, unexpected = model.load_state_dict({k: v.detach() for k, v in params.items()}, strict=True)
    assert not missing and not unexpected
    model.train()
    model.roi_heads.box_head.dropout_p = 0.0
    assert sorted(n for n, p in model.named_parameters() if p.requires_grad) == sorted(names)
    losses = model(to_batched_inputs(batch))
    sum(losses.values()).backward()
    assert set(losses) == set(losses0) and "loss_sem_seg" not in losses
    for k in sorted(losses0):
        a, b = float(losses[k].detach()), float(losses0[k])
        assert abs(a - b) <= loss_bar * max(abs(b), 1e-6) + 1e-7, (k, a, b)
    aux = model.roi_heads.aux
    cnt = aux["things_cnt"].cpu().tolist()
    for k in range(4):
        pad = aux["pgt_idx_r%d" % k].cpu().to(torch.int64)
        for i, b in enumerate(aux0["pgt_idx_r%d" % k]):
            assert cnt[i] == b.numel() and torch.equal(pad[i, :cnt[i]], b)
        assert torch.equal(aux["labels_r%d" % k].cpu().to(torch.int64), aux0["labels_r%d" % k])
    assert torch.equal(aux["fg_rois"].cpu(), aux0["fg_rois"]) and torch.equal(aux["fg_classes"].cpu(), aux0["fg_classes"])
    assert torch.equal(model.roi_heads.pgt_sem_seg.cpu(), aux0["sem_target"])
    a, b = aux["pooled_argmax"].cpu().contiguous(), aux0["pooled_argmax"]
    assert torch.equal(a == -1, b == -1) and (a != b).float().mean().item() < 2e-3
    got = dict(model.named_parameters())
    worst = {}
    for n in names:
        if n.endswith("box_predictor.det.bias"):
            continue   # exactly zero in exact arithmetic
        g = got[n].grad
        if n.endswith("box_head.fc1.weight"):
            g = model.roi_heads.box_head._hwc_cols(g, False)
        g0 = params[n].grad
        d = g.cpu().double() - g0.double()
        # (max-norm, L2) relative errors.  A ReLU gate of fc1 / fc2 or a max-pool winner that falls on the other side
        # in the two summation orders changes whole gradient rows (a discrete effect, see test_hip_model.py): the
        # max-norm bar is loose, the L2 bar — which a handful of flipped rows barely moves — is tight.
        worst[n] = (d.abs().max().item() / (g0.abs().max().item() + 1e-8), (d.norm() / (g0.double().norm() + 1e-12)).item())
    bad = {k: v for k, v in worst.items() if v[0] > 5e-2 or v[1] > 5e-3}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1][0])[:6]


def test_dc5_inference_contract(cuda):
    """The shipped configuration in eval mode: single-level MOIPool on the dilated res5 map (2048 channels: eight
    channel blocks in the pooling kernels), 20 classes, the parameter-free two-class semantic head -> detections, masks
    and a panoptic map whose only stuff segment is class 1."""
    torch.manual_seed(0)
    cfg = dc5_cfg("cuda")
    cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST = 1e-5
    cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST = 0.3
    cfg.MODEL.PANOPTIC_FPN.COMBINE.INSTANCES_CONFIDENCE_THRESH = 0.9     # keep most of the map for the stuff class
    model = build_model(cfg)
    with torch.no_grad():
        model.backbone.stem.conv1.weight.mul_(1.0 / 64)
    model.eval()
    inputs = synthetic_inputs(99, batch=2, size=256, proposals=120, sp_block=8, device=cuda, num_things=20, num_stuff=2,
                              n_stuff=1)
    out = model(inputs)
    assert len(out) == 2
    for o in out:
        inst, sem, (pan, info) = o["instances"], o["sem_seg"], o["panoptic_seg"]
        assert sem.shape == (2, 256, 256) and bool((sem[1] > sem[0]).all())
        assert len(inst) <= 100 and inst.pred_masks.shape[1:] == (256, 256) and int(inst.pred_classes.max()) < 20
        stuff = [s for s in info if not s["isthing"]]
        assert [s["category_id"] for s in stuff] == [1] and int((pan == stuff[0]["id"]).sum()) == stuff[0]["area"]
