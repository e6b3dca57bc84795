"""GPU suite: the whole R50-FPN JTSM training step on the HIP path (jtsm_amd, built through the
reference-style registry from configs/jtsm_R_50_FPN_1x.yaml) against the torch-CPU oracle
(oracle/model.py) on the same seeded weights and synthetic batch (reduced size: the oracle must finish
in seconds).  Checked: every loss (1e-4 relative), the integer artefacts (MOIPool argmax, mined
pseudo-GT rows, refinement labels, foreground set, pseudo semantic target) bit-exact, and gradients."""
import numpy as np
import pytest
import torch

from model_util import jtsm_cfg, to_batched_inputs
from oracle import model as OM

pytestmark = pytest.mark.gpu

from jtsm_amd.layers import conv as K  # noqa: E402
from jtsm_amd.modeling import build_model  # noqa: E402


def _rel(a, b, floor=1e-8):
    """max error relative to the tensor's magnitude; `floor` keeps gradients that are zero in exact
    arithmetic (e.g. the det bias: the softmax over proposals sums to one) from dividing noise by noise."""
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return (a - b).abs().max().item() / (b.abs().max().item() + floor)


@pytest.fixture(scope="module", params=["f32", "bf16x3"])
def step(cuda, request):
    """One training step in each contraction arithmetic: exact fp32 MFMA and split-bf16 (csrc/conv_x3.h)."""
    old = K.MATH
    K.set_math(request.param)
    try:
        yield _run_step() + (request.param,)
    finally:
        K.set_math(old)


def _run_step():
    torch.manual_seed(0)
    params = OM.init_params(seed=3, random_bn=True, input_gain=1.0 / 64)
    # 8-px superpixels: every >=16-px box owns at least one (no all-zero rois -> no exact score ties)
    batch = OM.synthetic_batch(1234, B=2, size=256, R=160, sp_block=8)
    names = OM.trainable_names(params)
    for n in names:
        params[n].requires_grad_(True)
    losses0, aux0 = OM.forward_losses(params, batch, return_aux=True)
    sum(losses0.values()).backward()

    model = build_model(jtsm_cfg("cuda"))
    missing, unexpected = model.load_state_dict({k: v.detach() for k, v in params.items()}, strict=True)
    assert not missing and not unexpected
    model.train()
    model.roi_heads.box_head.dropout_p = 0.0          # parity runs without dropout
    losses = model(to_batched_inputs(batch))
    sum(losses.values()).backward()
    return params, names, losses0, aux0, model, losses


def test_trainable_set_matches(step):
    params, names, _, _, model, _, _ = step
    mine = sorted(n for n, p in model.named_parameters() if p.requires_grad)
    assert mine == sorted(names)


def test_losses_match(step):
    _, _, losses0, _, _, losses, _ = step
    assert set(losses) == set(losses0)
    for k in sorted(losses0):
        a, b = float(losses[k]), float(losses0[k])
        assert abs(a - b) <= 1e-4 * max(abs(b), 1e-6) + 1e-7, (k, a, b)


def test_integer_artefacts_bit_exact(step):
    _, _, _, aux0, model, _, _ = step
    aux = model.roi_heads.aux
    # argmax of a max-pool is decided by near-ties between feature cells; the two backbones differ in
    # fp32 summation order, so a few winners may flip (exactness on identical inputs: test_hip_pooling)
    a, b = aux["pooled_argmax"].cpu().contiguous(), aux0["pooled_argmax"]
    assert torch.equal(a == -1, b == -1)
    assert (a != b).float().mean().item() < 2e-3
    cnt = aux["things_cnt"].cpu().tolist()      # padded (B, num_classes) winners; image i has cnt[i] present classes
    for k in range(4):
        pad = aux["pgt_idx_r%d" % k].cpu().to(torch.int64)
        for i, b in enumerate(aux0["pgt_idx_r%d" % k]):
            assert cnt[i] == b.numel() and torch.equal(pad[i, :cnt[i]], b)
        assert torch.equal(aux["labels_r%d" % k].cpu().to(torch.int64), aux0["labels_r%d" % k])
    assert torch.equal(aux["fg_rois"].cpu(), aux0["fg_rois"])
    assert torch.equal(aux["fg_classes"].cpu(), aux0["fg_classes"])
    assert torch.equal(model.roi_heads.pgt_sem_seg.cpu(), aux0["sem_target"])
    # mask pseudo labels: the "10 nearest" foreground proposals of every pseudo box (rows, in order), and the
    # superpixel-evidence targets of the first mask head — bit-exact; the refinery's paste -> crop targets come from
    # each side's own logits (1e-5 apart), so a pixel whose pasted probability sits on 0.5 may differ
    near = aux["near_rows"].cpu()
    for i, want in enumerate(aux0["near_rows"]):
        got = near[i, :cnt[i]].reshape(-1)
        assert torch.equal(got[got >= 0].to(torch.int64), want), (i, got, want)
    assert aux["mask_targets"].shape[0] > 0
    assert torch.equal(aux["mask_targets"].cpu(), aux0["mask_targets"])
    assert (aux["mask_targets_r0"].cpu() != aux0["mask_targets_r0"]).float().mean().item() < 2e-3


def test_rectangle_mask_targets_mode_still_matches(cuda):
    """MASK_TARGETS = "rect" (round 1's substitution: eroded pseudo-GT rectangles, thresholded refinery targets)
    stays available and stays in parity with the oracle's same mode."""
    params = OM.init_params(seed=3, random_bn=True, input_gain=1.0 / 64)
    batch = OM.synthetic_batch(1234, B=2, size=256, R=160, sp_block=8)
    losses0 = OM.forward_losses(params, batch, mask_targets="rect")
    model = build_model(jtsm_cfg("cuda"))
    model.load_state_dict({k: v.detach() for k, v in params.items()}, strict=True)
    model.train()
    model.roi_heads.box_head.dropout_p = 0.0
    model.roi_heads.mask_targets = "rect"
    losses = model(to_batched_inputs(batch))
    for k in ("loss_mask", "loss_mask_r0"):
        a, b = float(losses[k].detach()), float(losses0[k])
        assert abs(a - b) <= 1e-4 * max(abs(b), 1e-6) + 1e-7, (k, a, b)


def test_gradients_match(step):
    params, names, _, _, model, _, math = step
    got = dict(model.named_parameters())
    worst = {}
    for n in names:
        if n.endswith("box_predictor.det.bias"):
            continue  # exactly zero in exact arithmetic (softmax over the bag sums to 1): pure rounding noise
        g0, g = params[n].grad, got[n].grad
        assert g is not None, n
        if n.endswith("box_head.fc1.weight"):   # stored with (h,w,c) columns; the oracle keeps the reference's (c,h,w)
            g = model.roi_heads.box_head._hwc_cols(g, False)
        d = g.detach().cpu().double() - g0.double()
        worst[n] = (_rel(g, g0), (d.norm() / (g0.double().norm() + 1e-30)).item())
    # Free-running comparison: a handful of max-pool winners / ReLU gates fall on different sides in the two
    # implementations and change gradient rows outright (which ones depends on last-bit rounding, so the max-norm
    # error of an affected tensor is O(1e-2) and moves with any change of summation order).  The bars here say "no
    # more than a few such rows": relative L2 error 1e-2, max-norm 5e-2 (measured over the summation orders this
    # repo has had: L2 up to 5.4e-3 on res3.0, max-norm up to 1.3e-2).  The ARITHMETIC bar (1e-4 end to end) is
    # test_whole_step_gradients_at_1e4_with_frozen_discrete_choices, where those choices are taken out.
    bad = {k: v for k, v in worst.items() if v[0] > 5e-2 or v[1] > 1e-2}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1][0])[:8]


# The fp16 leg (BASELINE configs[4]): operands round at 2^-11 = 4.9e-4 where split-bf16 rounds at ~4e-6, and the MIL
# gradient amplifies relative errors by ~1e2 (the bf16x3 step's 5.6e-4 ... 7.5e-4 is that amplification of ~6e-6 per layer).
# Stated end-to-end gradient tolerance of the fp16 arithmetic with the discrete choices frozen (max-norm relative to
# each tensor's largest entry); measured: 3.2e-2 (det.weight), 1.5e-2 (fc2); the test prints the worst three.
FP16_GRAD_TOL = 5e-2
# fp16 operands round at 2^-11 (4.9e-4) relative; through ~105 convolutions and the heads the losses come out within a
# few 1e-3 of the fp32 oracle's.  Stated tolerance of the fp16 leg: 2e-2 relative per loss.
FP16_LOSS_TOL = 2e-2


@pytest.mark.parametrize("math,recipe", [("f32", "uniform"), ("bf16x3", "uniform"), ("f16", "uniform"),
                                         ("f32", "clustered"), ("bf16x3", "clustered")])
def test_whole_step_gradients_at_1e4_with_frozen_discrete_choices(cuda, math, recipe):
    """End-to-end gradients at the 1e-4 bar, in both arithmetics.  test_gradients_match has to accept ~1e-2 because a
    ReLU gate or a max-pool winner that sits within rounding of a tie falls on different sides in two correct
    implementations and changes gradient rows outright.  Here the discrete choices are FROZEN: the product runs first
    (layer by layer, so forward hooks see every convolution's output), its ReLU gates (y > 0), MOIPool winners and
    refinery mask targets are handed to the oracle (oracle/model.py `forced`), and both sides differentiate the same
    piecewise-linear map.  What remains is arithmetic: every trainable parameter's gradient within 1e-4 (max-norm,
    relative to the tensor's largest entry) with exact fp32 MFMA, 1e-3 with the split-bf16 contractions (see below).
    recipe "clustered" (VERDICT r3 item 9): the BENCHMARKED label path — proposals piled on 8 rectangles per image, so a
    pseudo box has dozens of foreground rois and the two mask towers' gradients carry weight in the step."""
    from jtsm_amd.layers import fused_blocks
    from jtsm_amd.layers.wrappers import Conv2d, ConvTranspose2d, Linear

    old_math, old_fused = K.MATH, fused_blocks.ENABLED
    K.set_math(math)
    fused_blocks.ENABLED = False
    try:
        torch.manual_seed(0)
        params = OM.init_params(seed=3, random_bn=True, input_gain=1.0 / 64)
        batch = (OM.synthetic_batch(1234, B=2, size=256, R=160, sp_block=8) if recipe == "uniform" else
                 OM.synthetic_batch(1234, B=2, size=256, R=160, sp_block=8, cluster=1.0, objects=8))
        names = OM.trainable_names(params)
        model = build_model(jtsm_cfg("cuda"))
        model.load_state_dict({k: v.detach() for k, v in params.items()}, strict=True)
        model.train()
        model.roi_heads.box_head.dropout_p = 0.0
        gates = {}

        def grab(name):
            def hook(mod, inp, out):
                gates[name] = (out.detach() > 0).cpu()
            return hook

        for name, mod in model.named_modules():
            relu = isinstance(mod, Conv2d) and mod.activation is not None
            if relu or isinstance(mod, ConvTranspose2d) or (isinstance(mod, Linear) and ".box_head.fc" in name):
                mod.register_forward_hook(grab(name))
        losses = model(to_batched_inputs(batch))
        sum(losses.values()).backward()
        aux = model.roi_heads.aux
        # the two mask heads share module objects per head; the stem / res2 gates matter for the forward values only
        forced = {"gates": gates, "argmax": aux["pooled_argmax"].cpu().contiguous(),
                  "mask_targets_r0": aux["mask_targets_r0"].cpu()}
        for n in names:
            params[n].requires_grad_(True)
        losses0 = OM.forward_losses(params, batch, forced=forced)
        sum(losses0.values()).backward()
        loss_tol = FP16_LOSS_TOL if math == "f16" else 1e-4
        for k in sorted(losses0):
            a, b = float(losses[k].detach()), float(losses0[k])
            assert abs(a - b) <= loss_tol * max(abs(b), 1e-6) + 1e-7, (k, a, b)
        got = dict(model.named_parameters())
        worst = {}
        for n in names:
            if n.endswith("box_predictor.det.bias"):
                continue  # exactly zero in exact arithmetic
            g0, g = params[n].grad, got[n].grad
            if n.endswith("box_head.fc1.weight"):
                g = model.roi_heads.box_head._hwc_cols(g, False)
            worst[n] = _rel(g, g0)
        top = sorted(worst.items(), key=lambda kv: -kv[1])[:3]
        print("frozen-choices gradient errors (%s, %s), worst three:" % (math, recipe), [(k, "%.2e" % v) for k, v in top])
        if recipe == "clustered":
            assert aux["fg_rois"].shape[0] >= 40            # the mask branch counts here
        # det.weight: the detection-stream gradient of every class column sums to zero over the bag (softmax over
        # proposals), so its entries are differences of nearly equal terms — measured 1.3e-4, bar 5e-4 for it alone
        # The split-bf16 contractions are ~20x less exact per layer than fp32 MFMA (6e-6 against 3e-7, both far inside the
        # 1e-4 bar per LAYER, tests/test_hip_conv.py); the MIL gradient is ill-conditioned (amplification ~1e2: the same
        # effect puts fp32's det.weight at 1.3e-4), so the END-TO-END gradients of the bf16x3 step agree to 5.6e-4 ... 7.5e-4
        # (measured in rounds 2 / 3; the DAN and predictor layers, which sit right behind the MIL loss) — bar 1e-3 for that
        # arithmetic; fp16: 3.2e-2 on det.weight, 1.5e-2 on the DAN (measured) — bar FP16_GRAD_TOL.
        lim = {"f32": 1e-4, "bf16x3": 1e-3, "f16": FP16_GRAD_TOL}[math]
        bad = {k: v for k, v in worst.items() if v > (max(lim, 5e-4) if k.endswith("box_predictor.det.weight") else lim)}
        assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]
    finally:
        K.set_math(old_math)
        fused_blocks.ENABLED = old_fused


def _check_against_oracle(model, losses, losses0, aux0):
    assert set(losses) == set(losses0)
    for k in sorted(losses0):
        a, b = float(losses[k].detach()), float(losses0[k])
        assert abs(a - b) <= 1e-4 * max(abs(b), 1e-6) + 1e-7, (k, a, b)
    aux = model.roi_heads.aux
    cnt = aux["things_cnt"].cpu().tolist()
    for k in range(4):
        pad = aux["pgt_idx_r%d" % k].cpu().to(torch.int64)
        for i, b in enumerate(aux0["pgt_idx_r%d" % k]):
            assert cnt[i] == b.numel() and torch.equal(pad[i, :cnt[i]], b)
        assert torch.equal(aux["labels_r%d" % k].cpu().to(torch.int64), aux0["labels_r%d" % k])
    assert torch.equal(aux["fg_rois"].cpu(), aux0["fg_rois"])
    assert torch.equal(aux["fg_classes"].cpu(), aux0["fg_classes"])
    assert torch.equal(model.roi_heads.pgt_sem_seg.cpu(), aux0["sem_target"])
    near = aux["near_rows"].cpu()
    for i, want in enumerate(aux0["near_rows"]):
        got = near[i, :cnt[i]].reshape(-1)
        assert torch.equal(got[got >= 0].to(torch.int64), want), (i, got, want)
    assert torch.equal(aux["mask_targets"].cpu(), aux0["mask_targets"])
    assert (aux["mask_targets_r0"].cpu() != aux0["mask_targets_r0"]).float().mean().item() < 2e-3
    return aux


@pytest.mark.parametrize("math", ["f32", "bf16x3"])
@pytest.mark.parametrize("objects", [5, 8])
def test_clustered_workload_matches_oracle(cuda, math, objects):
    """parity on the workload that is benchmarked.  bench.py's proposals are jittered copies of a few
    rectangles per image (cluster = 1): a mined pseudo box then has tens of foreground proposals with crowded, nearly
    equal IoUs — where the tie-breaking of the "10 nearest" targets (roi_heads_jtsm.py:840-905) and of the matcher's
    first maximum (matcher.py:61-103) decides integer artefacts.  Same recipe in oracle/model.py: every loss at 1e-4,
    every integer artefact bit-exact, in both arithmetics."""
    old = K.MATH
    K.set_math(math)
    try:
        params = OM.init_params(seed=3, random_bn=True, input_gain=1.0 / 64)
        batch = OM.synthetic_batch(1234, B=2, size=256, R=160, sp_block=8, cluster=1.0, objects=objects)
        losses0, aux0 = OM.forward_losses(params, batch, return_aux=True)
        assert aux0["fg_rois"].shape[0] >= 40              # the mask branch counts: dozens of rois per pseudo box
        assert all(n.numel() == 30 for n in aux0["near_rows"])   # 3 classes x 10 nearest, every list full
        model = build_model(jtsm_cfg("cuda"))
        model.load_state_dict({k: v.detach() for k, v in params.items()}, strict=True)
        model.train()
        model.roi_heads.box_head.dropout_p = 0.0
        losses = model(to_batched_inputs(batch))
        _check_against_oracle(model, losses, losses0, aux0)
        sum(losses.values()).backward()
        assert all(p.grad is None or bool(torch.isfinite(p.grad).all()) for p in model.parameters())
    finally:
        K.set_math(old)


def test_semantic_target_rect_mode_still_matches(cuda):
    """sem_targets = "rect" (round 1's substitution: pseudo-GT rectangles shrunk by 2 px) stays in parity with the
    oracle's same mode; the default paints the targets' superpixel-evidence masks as the reference does."""
    params = OM.init_params(seed=3, random_bn=True, input_gain=1.0 / 64)
    batch = OM.synthetic_batch(1234, B=2, size=256, R=160, sp_block=8)
    losses0, aux0 = OM.forward_losses(params, batch, sem_targets="rect", return_aux=True)
    _, aux1 = OM.forward_losses(params, batch, return_aux=True)
    assert not torch.equal(aux0["sem_target"], aux1["sem_target"])       # the two constructions differ
    model = build_model(jtsm_cfg("cuda"))
    model.load_state_dict({k: v.detach() for k, v in params.items()}, strict=True)
    model.train()
    model.roi_heads.box_head.dropout_p = 0.0
    model.roi_heads.sem_targets = "rect"
    losses = model(to_batched_inputs(batch))
    assert torch.equal(model.roi_heads.pgt_sem_seg.cpu(), aux0["sem_target"])
    a, b = float(losses["loss_sem_seg"].detach()), float(losses0["loss_sem_seg"])
    assert abs(a - b) <= 1e-4 * abs(b)


def test_semantic_target_from_evidence_masks_bit_exact(cuda):
    """jtsm_paint_sem_seg_evidence against the oracle's literal reading of get_pgt_sem_seg (roi_heads_jtsm.py:2038-2069
    with the masks of :1928-1994) on crafted targets: overlapping evidence painted in ascending-score order, a class
    painted over COMPLETELY (repainted by the second pass, in list order), equal scores (lower index first — argsort is
    stable in the oracle's torch build for this size), superpixel ids outside [0, L), a ragged target count, and the
    reference-shaped list-of-Instances entry."""
    from jtsm_amd.layers.mining import paint_sem_seg_evidence

    g = torch.Generator().manual_seed(11)
    B, H, W, L, R, G = 2, 48, 64, 24, 12, 5
    sp = torch.randint(-1, L + 2, (B, H, W), generator=g, dtype=torch.int32)      # ids -1, L, L+1: no mask owns them
    oh = [(torch.rand(R, L, generator=g) < 0.3).to(torch.int32) for _ in range(B)]
    idx = torch.tensor([[3, 5, 7, 1, 0], [2, 9, 4, 0, 0]])
    counts = [5, 3]
    scores = torch.tensor([[0.2, 0.9, 0.2, 0.5, 0.7], [0.6, 0.1, 0.6, 0.0, 0.0]])
    classes = torch.tensor([[81, 84, 90, 95, 100], [82, 83, 99, 0, 0]])
    # image 0: target 1 has the top score and its evidence covers target 0's -> class 81 is painted over and comes
    # back in the second pass; target 3 marks nothing at all (stays absent)
    oh[0][5] = torch.maximum(oh[0][5], oh[0][3])
    oh[0][1] = 0
    tg = [dict(idx=idx[i, :counts[i]], classes=classes[i, :counts[i]].to(torch.int64), scores=scores[i, :counts[i]])
          for i in range(B)]
    want = OM.pgt_sem_seg(tg, H, W, 80, oh, sp)
    assert (want[0] == 2).any() and not (want[0] == 16).any()
    offsets = torch.tensor([0, R, 2 * R], dtype=torch.int32, device=cuda)
    got = paint_sem_seg_evidence(idx.to(cuda), offsets, torch.cat(oh).to(cuda), sp.to(cuda), classes.to(cuda),
                                 scores.to(cuda), torch.tensor(counts, device=cuda), 79)
    assert got.dtype == torch.int64 and torch.equal(got.cpu(), want)


def test_full_size_clustered_step_is_reproducible_and_finite(cuda):
    """The benchmarked step itself (bench.py defaults: cluster = 1, 40 rectangles per image, ~300 foreground rois):
    finite losses and gradients, reproducible losses, and the label path's invariants at full size — every foreground
    row carries a thing label, the near targets are foreground rows of the right image, every mask target row exists."""
    from jtsm_amd.utils.synthetic import synthetic_inputs

    torch.manual_seed(0)
    model = build_model(jtsm_cfg("cuda"))
    model.train()
    model.roi_heads.box_head.dropout_p = 0.0
    with torch.no_grad():
        model.backbone.bottom_up.stem.conv1.weight.mul_(1.0 / 64)
    inputs = synthetic_inputs(1234, batch=2, size=1024, proposals=2000, device=cuda, cluster=1.0, objects=40)
    runs = []
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        losses = model(inputs)
        sum(losses.values()).backward()
        runs.append(({k: float(v) for k, v in losses.items()},
                     {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.requires_grad}))
    (l0, g0), (l1, g1) = runs
    assert all(np.isfinite(v) for v in l0.values()), l0
    for k in l0:
        assert abs(l0[k] - l1[k]) <= 1e-6 * max(abs(l0[k]), 1e-6), (k, l0[k], l1[k])
    for n in g0:
        assert bool(torch.isfinite(g0[n]).all()), n
        if n.endswith("box_predictor.det.bias"):
            continue
        assert float((g0[n] - g1[n]).abs().max()) <= 1e-4 * (float(g0[n].abs().max()) + 1e-12), n
    aux = model.roi_heads.aux
    fg = aux["fg_rois"]
    assert fg.shape[0] >= 100 and aux["mask_targets"].shape[0] == fg.shape[0]
    assert bool(((aux["fg_classes"] >= 0) & (aux["fg_classes"] < 80)).all())
    labels = aux["labels_r3"]
    near = aux["near_rows"]                               # (B, G, 10) global rows, -1 = empty slot
    cnt = aux["things_cnt"].cpu().tolist()
    for b in range(2):
        rows = near[b, :cnt[b]].reshape(-1)
        rows = rows[rows >= 0].to(torch.int64)
        assert rows.numel() > 0 and bool(((rows >= 2000 * b) & (rows < 2000 * (b + 1))).all())
    sem = model.roi_heads.pgt_sem_seg
    assert sem.shape == (2, 1024, 1024) and int(sem.max()) <= 53 and int(sem.min()) >= 0
    assert labels.shape[0] == 4000


def test_side_streams_give_reproducible_training_trajectories(cuda):
    """The benchmarked step with its side streams (semantic head beside the box / mask branches, weight gradients beside
    the data gradients): six optimizer steps from the same weights, three times — the trajectories are identical bit for
    bit among themselves; without the semantic side stream they are identical to the ONE-stream trajectory; with it they
    differ from that only by the pyramid gradients' summation order (last bits).  Round 4's side-stream anomaly (DESIGN
    §5; the up-sampling kernel built with packed fp32 arithmetic) made ~9 of 10 such forward passes differ."""
    import copy
    from jtsm_amd.layers import conv as K
    from jtsm_amd.modeling.meta_arch import mcnn
    from jtsm_amd.utils.synthetic import synthetic_inputs

    torch.manual_seed(0)
    model = build_model(jtsm_cfg("cuda"))
    model.train()
    with torch.no_grad():
        model.backbone.bottom_up.stem.conv1.weight.mul_(1.0 / 64)
    inputs = synthetic_inputs(1234, batch=2, size=1024, proposals=2000, device=cuda, cluster=1.0, objects=40)
    init = copy.deepcopy(model.state_dict())
    params = [p for p in model.parameters() if p.requires_grad]
    keep = (mcnn.SEM_SIDE_STREAM, K.WGRAD_STREAM)

    def trajectory(sem_side, wgrad_side, steps=6):
        mcnn.SEM_SIDE_STREAM, K.WGRAD_STREAM = sem_side, wgrad_side
        model.load_state_dict(init)
        opt = torch.optim.SGD(params, lr=1e-3, momentum=0.9)
        torch.manual_seed(7)                                   # (the dropout seeds of the box head)
        out = []
        for _ in range(steps):
            losses = model(inputs)
            sum(losses.values()).backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
            out.append(torch.stack([losses[k].detach() for k in sorted(losses)]))
        torch.cuda.synchronize()
        return torch.stack(out), torch.cat([p.detach().flatten() for p in params[:8]])

    try:
        ref, ref_w = trajectory(False, False)
        assert bool(torch.isfinite(ref).all())
        t, w = trajectory(False, True)
        assert torch.equal(t, ref) and torch.equal(w, ref_w), "weight-gradient side stream changed the trajectory"
        first, first_w = trajectory(True, True)
        # (same forward: the first step's losses are the one-stream bits; later steps carry the other summation order
        # through a steep trajectory — the semantic loss falls from 15 to 6.5 in these six steps)
        assert torch.equal(first[0], ref[0]), (first[0] - ref[0]).abs().max()
        assert float((first[1] - ref[1]).abs().max()) <= 1e-5 * float(ref[1].abs().max()), (first[1] - ref[1]).abs().max()
        assert float((first - ref).abs().max()) <= 1e-2 * float(ref.abs().max()), (first - ref).abs().max()
        for _ in range(2):
            t, w = trajectory(True, True)
            assert torch.equal(t, first) and torch.equal(w, first_w), "side streams: trajectory not reproducible"
    finally:
        mcnn.SEM_SIDE_STREAM, K.WGRAD_STREAM = keep


def test_early_heads_update_is_the_plain_update(cuda):
    """solver/build.py: attach_early_heads — the heads' parameters are updated (and their operand planes refreshed) on the
    weight-gradient side stream when the backward reaches the FPN, the rest in step(): four steps from the same weights
    must end in the same bits as the plain fused step, and the early part must really have run (two update launches per
    step)."""
    import copy
    from jtsm_amd import _lib as L
    from jtsm_amd.solver import SGD
    from jtsm_amd.utils.synthetic import synthetic_inputs

    torch.manual_seed(0)
    model = build_model(jtsm_cfg("cuda"))
    model.train()
    with torch.no_grad():
        model.backbone.bottom_up.stem.conv1.weight.mul_(1.0 / 64)
    inputs = synthetic_inputs(1234, batch=2, size=512, proposals=400, device=cuda, cluster=1.0, objects=12)
    init = copy.deepcopy(model.state_dict())
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]

    def run(early):
        model.load_state_dict(init)
        decay = [p for n, p in named if not n.endswith(".bias")]
        bias = [p for n, p in named if n.endswith(".bias")]
        opt = SGD([{"params": decay, "lr": 1e-4, "weight_decay": 5e-4}, {"params": bias, "lr": 2e-4, "weight_decay": 0.0}],
                  lr=1e-4, momentum=0.9)
        if early:
            opt.attach_early_heads(model)
            assert opt._early_ids
        torch.manual_seed(7)
        launches = 0
        try:
            for it in range(4):
                losses = model(inputs)
                L.TIMING = []
                sum(losses.values()).backward()
                opt.step()
                launches = sum(1 for name, _, _ in L.TIMING if name == "jtsm_sgd_momentum_multi_f32")
                L.TIMING = None
                opt.zero_grad(set_to_none=True)
        finally:
            L.TIMING = None
            if early:
                opt.detach_early_heads()
        torch.cuda.synchronize()
        return {n: p.detach().clone() for n, p in named}, launches

    plain, n_plain = run(False)
    early, n_early = run(True)
    assert n_plain == 1 and n_early == 2, (n_plain, n_early)
    for n in plain:
        assert torch.equal(plain[n], early[n]), n


def test_full_size_step_is_reproducible_and_finite(cuda):
    """BASELINE configs[2] at full size (2 x 1024^2, 2000 proposals per image): too big for the CPU oracle, so check
    what needs none — every loss finite, every trainable parameter gets a finite gradient, and a second run of the
    same step reproduces the losses (the bf16x3 contractions and all reductions are deterministic; only the pooling
    backward's float atomics may move last bits of the gradients)."""
    from jtsm_amd.utils.synthetic import synthetic_inputs

    torch.manual_seed(0)
    model = build_model(jtsm_cfg("cuda"))
    model.train()
    model.roi_heads.box_head.dropout_p = 0.0
    with torch.no_grad():
        model.backbone.bottom_up.stem.conv1.weight.mul_(1.0 / 64)
    inputs = synthetic_inputs(1234, batch=2, size=1024, proposals=2000, device=cuda)
    runs = []
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        losses = model(inputs)
        sum(losses.values()).backward()
        runs.append(({k: float(v) for k, v in losses.items()},
                     {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.requires_grad}))
    (l0, g0), (l1, g1) = runs
    assert all(np.isfinite(v) for v in l0.values()), l0
    for k in l0:
        assert abs(l0[k] - l1[k]) <= 1e-6 * max(abs(l0[k]), 1e-6), (k, l0[k], l1[k])
    for n in g0:
        assert g0[n] is not None and bool(torch.isfinite(g0[n]).all()), n
        if n.endswith("box_predictor.det.bias"):
            continue  # exactly zero in exact arithmetic (see test_gradients_match): rounding noise only
        ref = float(g0[n].abs().max()) + 1e-12
        assert float((g0[n] - g1[n]).abs().max()) <= 1e-4 * ref, n


def test_ragged_batch_losses_and_labels(cuda):
    """Images with different proposal counts and different numbers of present thing / stuff classes (the padded
    device-side class lists and the per-image bag offsets must cope): losses and integer artefacts vs the oracle."""
    params = OM.init_params(seed=5, random_bn=True, input_gain=1.0 / 64)
    batch = OM.synthetic_batch(4321, B=2, size=256, R=160, sp_block=8)
    keep = 97                                            # image 1 keeps 97 of its 160 proposals
    for k in ("boxes", "objectness", "oh_labels"):
        batch[k][1] = batch[k][1][:keep].contiguous()
    batch["gt_classes"][1] = batch["gt_classes"][1][:2].contiguous()     # 3 thing classes vs 2
    sem = batch["sem_seg"]
    stuff1 = [int(v) for v in torch.unique(sem[1]) if int(v) not in (0, 255)]
    sem[1][sem[1] == stuff1[-1]] = 0                                      # 2 stuff classes vs 1
    losses0, aux0 = OM.forward_losses(params, batch, return_aux=True)

    model = build_model(jtsm_cfg("cuda"))
    model.load_state_dict({k: v.detach() for k, v in params.items()}, strict=True)
    model.train()
    model.roi_heads.box_head.dropout_p = 0.0
    losses = model(to_batched_inputs(batch))
    assert set(losses) == set(losses0)
    for k in sorted(losses0):
        a, b = float(losses[k].detach()), float(losses0[k])
        assert abs(a - b) <= 1e-4 * max(abs(b), 1e-6) + 1e-7, (k, a, b)
    aux = model.roi_heads.aux
    cnt = aux["things_cnt"].cpu().tolist()
    assert cnt == [3, 2]
    for k in range(4):
        pad = aux["pgt_idx_r%d" % k].cpu().to(torch.int64)
        for i, b in enumerate(aux0["pgt_idx_r%d" % k]):
            assert torch.equal(pad[i, :cnt[i]], b)
        assert torch.equal(aux["labels_r%d" % k].cpu().to(torch.int64), aux0["labels_r%d" % k])
    assert torch.equal(aux["fg_rois"].cpu(), aux0["fg_rois"])
    assert torch.equal(model.roi_heads.pgt_sem_seg.cpu(), aux0["sem_target"])


def _r101_rect_case():
    params = OM.init_params(seed=7, depth=101, random_bn=True, input_gain=1.0 / 64)
    with torch.no_grad():   # 33 random-init residual blocks double the variance each: damp every block's last norm
        for k in params:
            if k.endswith("conv3.norm.weight"):
                params[k] *= 0.3
    batch = OM.synthetic_batch(99, B=1, size=256, R=96, sp_block=8)
    # crop to 128 x 256 (H x W): boxes / labels / superpixels follow
    H, W = 128, 256
    batch["images"] = [im[:, :H, :W].contiguous() for im in batch["images"]]
    batch["sem_seg"] = batch["sem_seg"][:, :H, :W].contiguous()
    batch["superpixels"] = batch["superpixels"][:, :H, :W].contiguous()
    bx = batch["boxes"][0].clone()
    bx[:, 1] *= 0.5
    bx[:, 3] *= 0.5                                       # the same boxes squeezed into the top half
    batch["boxes"][0] = bx
    grid = 256 // 8
    cy = torch.arange(grid) * 8 + 4.0
    iny = (cy[None, :] >= bx[:, 1:2]) & (cy[None, :] <= bx[:, 3:4])
    inx = (cy[None, :] >= bx[:, 0:1]) & (cy[None, :] <= bx[:, 2:3])
    batch["oh_labels"][0] = (iny[:, :, None] & inx[:, None, :]).reshape(len(bx), -1).to(torch.int32)
    losses0 = OM.forward_losses(params, batch, depth=101)
    return params, batch, losses0


def _r101_rect_step(params, batch):
    model = build_model(jtsm_cfg("cuda", depth=101))
    model.load_state_dict({k: v.detach() for k, v in params.items()}, strict=True)
    model.train()
    model.roi_heads.box_head.dropout_p = 0.0
    losses = model(to_batched_inputs(batch))
    sum(losses.values()).backward()
    assert all(p.grad is None or bool(torch.isfinite(p.grad).all()) for p in model.parameters())
    return {k: float(v.detach()) for k, v in losses.items()}


def test_r101_rectangular_losses_match(cuda):
    """BASELINE configs[4] geometry at reduced size: R101-FPN on 1:2 (Cityscapes-shaped) images — every loss against
    the oracle at the fp32 bar (1e-4), in the default split-bf16 arithmetic."""
    params, batch, losses0 = _r101_rect_case()
    losses = _r101_rect_step(params, batch)
    assert set(losses) == set(losses0)
    for k in sorted(losses0):
        a, b = losses[k], float(losses0[k])
        assert abs(a - b) <= 1e-4 * max(abs(b), 1e-6) + 1e-7, (k, a, b)
    assert float(losses0["loss_mask"]) > 0                # the case has foreground rois



def test_r101_rectangular_fp16_losses_within_stated_tolerance(cuda):
    """BASELINE configs[4] arithmetic ("fp16 MFMA path": one fp16 plane per operand, v_mfma_f32_32x32x16_f16, fp32
    accumulate, fp32 losses) on the same reduced R101-FPN case, against the fp32 oracle at the stated fp16 tolerance."""
    params, batch, losses0 = _r101_rect_case()
    old = K.MATH
    K.set_math("f16")
    try:
        losses = _r101_rect_step(params, batch)
    finally:
        K.set_math(old)
    assert set(losses) == set(losses0)
    worst = {k: abs(losses[k] - float(losses0[k])) / max(abs(float(losses0[k])), 1e-6) for k in losses0}
    print("fp16 loss errors:", {k: "%.2e" % v for k, v in sorted(worst.items())})
    bad = {k: v for k, v in worst.items() if v > FP16_LOSS_TOL}
    assert not bad, bad


def test_fp16_only_identity_chain_matches_the_per_block_nodes(cuda):
    """fp16-only activations.  A stage's identity blocks as ONE node whose activations and gradient
    stream are fp16 planes alone (layers/fused_blocks.py: _IdentityChain16Fn; residual / shortcut gradient read from
    planes: jtsm_conv2d_forward_res16_f16 / jtsm_conv2d_backward_data_acc16_f16) against (1) the per-block nodes of the
    same fp16 arithmetic, which keep an fp32 residual stream, and (2) the same stage in exact fp32: output, input
    gradient and every weight gradient within the fp16 leg's stated bars (operands round at 2^-11 per layer either
    way; the chain also rounds the residual stream once per block, as the reference's AMP step does)."""
    from jtsm_amd.layers import fused_blocks
    from jtsm_amd.modeling.backbone.resnet import BottleneckBlock, ResNet

    torch.manual_seed(5)
    blocks = ResNet.make_stage(BottleneckBlock, 5, 2, in_channels=128, out_channels=256, norm="FrozenBN",
                               bottleneck_channels=64, stride_in_1x1=True)
    stage = torch.nn.Sequential(*blocks).to(cuda)
    with torch.no_grad():
        for n, b in stage.named_buffers():
            if n.endswith("norm.weight"):
                b.copy_(torch.rand_like(b) * 0.5 + (0.2 if "conv3" in n else 0.8))
            if n.endswith("norm.bias"):
                b.copy_(torch.randn_like(b) * 0.1)
    x0 = torch.relu(torch.randn(2, 128, 40, 48, device=cuda)).contiguous(memory_format=torch.channels_last)
    g0 = torch.randn(2, 256, 20, 24, device=cuda).contiguous(memory_format=torch.channels_last)
    weights = [p for p in stage.parameters()]

    def run(math, chain):
        old_math, old_chain = K.MATH, fused_blocks.CHAIN16
        K.set_math(math)
        fused_blocks.CHAIN16 = chain
        K.planes_clear()
        try:
            x = x0.clone().requires_grad_(True)
            for p in weights:
                p.grad = None
            y = ResNet._run_stage(stage, x)
            y.backward(g0)
            K.flush_deferred_weight_gradients()
            torch.cuda.synchronize()
            return y.detach().clone(), x.grad.clone(), [p.grad.clone() for p in weights]
        finally:
            K.set_math(old_math)
            fused_blocks.CHAIN16 = old_chain

    fused_blocks_nodes = run("f16", False)
    chain = run("f16", True)
    exact = run("f32", False)
    assert fused_blocks.identity_chain_ok(x0, list(stage.children())[1:]) is False     # (outside the fp16 arithmetic: off)

    def l2(a, b):
        a, b = a.double(), b.double()
        return float((a - b).norm() / (b.norm() + 1e-30))

    def errors(a, b, fn):
        return [fn(a[1], b[1])] + [fn(p, q) for p, q in zip(a[2], b[2])]
    # A ReLU gate within fp16 rounding of zero falls on different sides in two arithmetics and changes gradient entries
    # outright (tests above: the free-running gradient test), so the max-norm is compared between the two fp16 forms
    # and the arithmetic bar is applied to the L2 error
    max_chain, max_nodes = max(errors(chain, exact, _rel)), max(errors(fused_blocks_nodes, exact, _rel))
    l2_chain, l2_nodes = max(errors(chain, exact, l2)), max(errors(fused_blocks_nodes, exact, l2))
    print("fp16-only chain vs exact fp32: output %.2e, gradients max-norm %.2e / L2 %.2e (per-block fp16 nodes: %.2e / %.2e)"
          % (_rel(chain[0], exact[0]), max_chain, l2_chain, max_nodes, l2_nodes))
    assert _rel(chain[0], exact[0]) <= FP16_LOSS_TOL, _rel(chain[0], exact[0])          # activations: the loss bar
    assert l2_chain <= 2 * FP16_GRAD_TOL, errors(chain, exact, l2)                      # (gate flips included: see above)
    assert max_chain <= 1.5 * max_nodes + 1e-2 and l2_chain <= 1.5 * l2_nodes + 1e-3    # no worse than the fp32-stream form
    assert all(float(g.abs().max()) > 0 for g in chain[2])


def test_config4_full_size_fp16_step(cuda):
    """BASELINE configs[4] at FULL size on one GPU: R101-FPN JTSM panoptic, 2 x 3 x 1024 x 2048 (Cityscapes-shaped),
    2000 proposals per image (the configs[2] recipe scaled x2 in x), fp16 MFMA path.  Too big for the CPU oracle, so
    check what needs none: every loss finite, every trainable parameter gets a finite gradient, a second run of the
    same step reproduces the losses, and the losses agree with the SAME step in the fp32-parity arithmetic
    (split-bf16) within the stated fp16 tolerance."""
    from jtsm_amd.utils.synthetic import synthetic_inputs

    def build():
        torch.manual_seed(0)
        m = build_model(jtsm_cfg("cuda", depth=101))
        m.train()
        m.roi_heads.box_head.dropout_p = 0.0
        with torch.no_grad():
            m.backbone.bottom_up.stem.conv1.weight.mul_(1.0 / 64)
            for n, p in m.named_buffers():   # 33 random-init residual blocks: damp every block's last norm
                if n.endswith("conv3.norm.weight"):
                    p.mul_(0.3)
        return m

    inputs = synthetic_inputs(1234, batch=2, size=1024, width=2048, proposals=2000, device=cuda)
    old = K.MATH
    out = {}
    try:
        for math in ("f16", "bf16x3"):
            K.set_math(math)
            model = build()
            runs = []
            for _ in range(2 if math == "f16" else 1):
                model.zero_grad(set_to_none=True)
                losses = model(inputs)
                sum(losses.values()).backward()
                runs.append({k: float(v) for k, v in losses.items()})
            grads = {n: p.grad for n, p in model.named_parameters() if p.requires_grad}
            assert all(g is not None and bool(torch.isfinite(g).all()) for g in grads.values())
            out[math] = runs
            del model, grads
            torch.cuda.empty_cache()
    finally:
        K.set_math(old)
    l0, l1 = out["f16"]
    assert all(np.isfinite(v) for v in l0.values()), l0
    for k in l0:
        assert abs(l0[k] - l1[k]) <= 1e-6 * max(abs(l0[k]), 1e-6), (k, l0[k], l1[k])
    ref = out["bf16x3"][0]
    worst = {k: abs(l0[k] - ref[k]) / max(abs(ref[k]), 1e-6) for k in ref}
    print("configs[4] full size, fp16 vs bf16x3 losses:", {k: "%.2e" % v for k, v in sorted(worst.items())})
    assert max(worst.values()) <= FP16_LOSS_TOL, worst
