"""GPU suite: fused MIL / OICR loss kernels vs the torch-CPU oracle (oracle/model.py) — values and
gradients, ragged bags, strided (fused-GEMM) inputs, ignored rows.  Bar: 1e-4 relative."""
import pytest
import torch
import torch.nn.functional as F

from oracle import model as OM

pytestmark = pytest.mark.gpu

from jtsm_amd.layers.wsl_losses import mil_loss, oicr_loss  # noqa: E402


def rel_close(a, b, tol=1e-4, what=""):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    ref = b.abs().max().item() + 1e-30
    err = (a - b).abs().max().item()
    assert err <= tol * ref, "%s: err %.3e ref %.3e" % (what, err, ref)


@pytest.mark.parametrize("counts", [[2000, 2000], [37, 5, 120], [1], [700, 0, 3]])
@pytest.mark.parametrize("mean_loss", [True, False])
def test_mil_forward_backward(cuda, counts, mean_loss):
    g = torch.Generator().manual_seed(7)
    R, nc, B = sum(counts), OM.NUM_MIL, len(counts)
    wide = torch.randn(R, 2 * nc + 30, generator=g) * 2        # C and D are column ranges of one matrix
    y = (torch.rand(B, nc, generator=g) < 0.05).float()
    Cc, Dc = wide[:, :nc].clone().requires_grad_(), wide[:, nc:2 * nc].clone().requires_grad_()
    s0 = OM.mil_scores(Cc, Dc, counts)
    p0 = OM.mil_image_probs(s0, counts)
    l0 = F.binary_cross_entropy(p0, y, reduction="mean" if mean_loss else "sum")
    if not mean_loss:
        l0 = l0 / B
    (l0 * 1.7).backward()
    wd = wide.to(cuda).requires_grad_()
    off = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32, device=cuda)
    loss, scores, probs = mil_loss(wd[:, :nc], wd[:, nc:2 * nc], off, y.to(cuda), mean_loss, max(counts))
    rel_close(scores, s0, what="scores")
    rel_close(probs, p0, what="probs")
    rel_close(loss, l0, what="loss")
    (loss * 1.7).backward()
    rel_close(wd.grad[:, :nc], Cc.grad, what="dC")
    rel_close(wd.grad[:, nc:2 * nc], Dc.grad, what="dD")
    assert wd.grad[:, 2 * nc:].abs().max().item() == 0


@pytest.mark.parametrize("mean_loss", [True, False])
def test_tsm_output_layers_reference_shaped_surface(cuda, mean_loss):
    """TSMOutputLayers.forward(x, proposals) -> (scores, zero deltas), .losses(predictions, proposals, oh),
    .predict_probs_img / predict_probs / predict_boxes (fast_rcnn_tsm.py:548-598,672-694,790-854): values and the
    gradients that reach the cls / det Linears, against the oracle's literal per-image softmax chain; and the same
    loss as the fused score_and_loss path JTSMROIHeads uses."""
    from jtsm_amd.modeling.roi_heads.fast_rcnn_tsm import TSMOutputLayers
    from jtsm_amd.structures import Boxes, Instances

    g = torch.Generator().manual_seed(17)
    counts, feat = [37, 5, 90], 64
    R, B = sum(counts), len(counts)
    head = TSMOutputLayers(feat, num_classes=OM.NUM_THINGS, num_classes_stuff=OM.NUM_STUFF, mean_loss=mean_loss,
                           box2box_transform=__import__("jtsm_amd.modeling.box_regression", fromlist=["x"]).Box2BoxTransform(
                               weights=(10.0, 10.0, 5.0, 5.0))).to(cuda)
    x = torch.randn(R, feat, generator=g)
    y = (torch.rand(B, OM.NUM_MIL, generator=g) < 0.05).float()
    boxes = torch.rand(R, 4, generator=g) * 100
    boxes[:, 2:] += boxes[:, :2] + 4
    props = [Instances((256, 256), proposal_boxes=Boxes(b.to(cuda))) for b in boxes.split(counts)]
    # oracle chain on copies of the same weights
    wc, bc = head.cls.weight.detach().cpu().clone().requires_grad_(), head.cls.bias.detach().cpu().clone().requires_grad_()
    wd, bd = head.det.weight.detach().cpu().clone().requires_grad_(), head.det.bias.detach().cpu().clone().requires_grad_()
    s0 = OM.mil_scores(F.linear(x, wc, bc), F.linear(x, wd, bd), counts)
    p0 = OM.mil_image_probs(s0, counts)
    l0 = F.binary_cross_entropy(p0, y, reduction="mean" if mean_loss else "sum") / (1 if mean_loss else B)
    l0.backward()
    pred = head(x.to(cuda), props)
    assert pred[1].shape == (R, 4 * OM.NUM_MIL) and float(pred[1].abs().sum()) == 0.0
    rel_close(pred[0], s0, what="scores")
    rel_close(head.predict_probs_img(pred, props), p0, what="image probabilities")
    loss = head.losses(pred, props, y.to(cuda))["loss_cls"]
    rel_close(loss, l0, what="loss")
    loss.backward()
    for mine, ref, what in ((head.cls.weight, wc, "d cls.weight"), (head.cls.bias, bc, "d cls.bias"),
                            (head.det.weight, wd, "d det.weight")):
        rel_close(mine.grad, ref.grad, what=what)
    probs = head.predict_probs(pred, props)
    assert [tuple(t.shape) for t in probs] == [(n, OM.NUM_MIL + 1) for n in counts] and float(probs[0][:, -1].sum()) == 0
    pb = head.predict_boxes(pred, props)
    assert torch.allclose(pb[1][:, :4].cpu(), boxes.split(counts)[1], atol=1e-4)      # zero deltas: the boxes themselves
    # the fused path gives the same loss
    c, d = head.logits(x.to(cuda))
    off = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32, device=cuda)
    fused, _, _ = head.score_and_loss(c, d, off, y.to(cuda), max(counts))
    rel_close(fused["loss_cls"], loss, what="fused == reference-shaped")


@pytest.mark.parametrize("R", [4000, 77, 1])
@pytest.mark.parametrize("with_box", [True, False])
def test_oicr_forward_backward(cuda, R, with_box):
    g = torch.Generator().manual_seed(8)
    K = OM.NUM_THINGS
    z = (torch.randn(R, K + 1, generator=g) * 2).requires_grad_()
    d = (torch.randn(R, 4 * K, generator=g) * 0.5).requires_grad_()
    labels = torch.randint(0, K + 1, (R,), generator=g)
    labels[torch.rand(R, generator=g) < 0.7] = K
    if R > 3:
        labels[:3] = torch.tensor([-1, 0, K - 1])
    w = torch.rand(R, generator=g)
    w[torch.rand(R, generator=g) < 0.2] = 0.0
    if R == 1:
        w[:] = 0.5
    prop = torch.rand(R, 4, generator=g) * 200
    prop[:, 2:] += prop[:, :2] + 4
    gt = prop + torch.randn(R, 4, generator=g) * 3
    gt[:, 2:] = torch.max(gt[:, 2:], gt[:, :2] + 2)
    lab = dict(classes=labels, boxes=gt, weights=w)
    lc0, lb0 = OM.oicr_losses(z, d, prop, lab)
    (lc0 * 0.7 + (lb0 * 1.3 if with_box else 0)).backward()
    zd, dd = z.detach().to(cuda).requires_grad_(), d.detach().to(cuda).requires_grad_()
    lc, lb = oicr_loss(zd, dd if with_box else None, labels.to(cuda), w.to(cuda),
                       prop.to(cuda) if with_box else None, gt.to(cuda) if with_box else None)
    rel_close(lc, lc0, what="loss_cls")
    if with_box:
        rel_close(lb, lb0, what="loss_box")
    (lc * 0.7 + (lb * 1.3 if with_box else 0)).backward()
    rel_close(zd.grad, z.grad, what="dz")
    if with_box:
        rel_close(dd.grad, d.grad, what="dd")


@pytest.mark.parametrize("reg", [True, False])
def test_oicr_output_layers_reference_shaped_surface(cuda, reg):
    """OICROutputLayers.forward(x) -> (scores, deltas) and .losses(predictions, proposals) with a list[Instances]
    carrying proposal_boxes / gt_classes / gt_boxes / gt_weights (fast_rcnn_oicr.py:534-586, OICROutputs :180-380):
    values and the gradients that reach cls_score / bbox_pred against the oracle's weighted CE + weighted L1 on
    copies of the same weights; an image WITHOUT gt_boxes (all background: the proposal boxes stand in, :232-236);
    the same numbers as the fused tensor form JTSMROIHeads calls; an empty proposal list."""
    from jtsm_amd.modeling.box_regression import Box2BoxTransform
    from jtsm_amd.modeling.roi_heads.fast_rcnn_oicr import OICROutputLayers
    from jtsm_amd.structures import Boxes, Instances

    g = torch.Generator().manual_seed(23)
    counts, feat, K = [61, 9, 130], 64, OM.NUM_THINGS
    R = sum(counts)
    head = OICROutputLayers(feat, num_classes=K, box2box_transform=Box2BoxTransform(weights=(10.0, 10.0, 5.0, 5.0)),
                            refine_k=1, refine_reg=[False, reg, False, False]).to(cuda)
    with torch.no_grad():
        head.cls_score.weight.mul_(30)
        head.bbox_pred.weight.mul_(100)
    x = torch.randn(R, feat, generator=g)
    labels = torch.randint(0, K + 1, (R,), generator=g)
    labels[torch.rand(R, generator=g) < 0.6] = K
    labels[:2] = torch.tensor([-1, 3])
    w = torch.rand(R, generator=g)
    w[torch.rand(R, generator=g) < 0.2] = 0.0
    prop = torch.rand(R, 4, generator=g) * 200
    prop[:, 2:] += prop[:, :2] + 4
    gt = prop + torch.randn(R, 4, generator=g) * 3
    gt[:, 2:] = torch.max(gt[:, 2:], gt[:, :2] + 2)
    s1 = slice(counts[0], counts[0] + counts[1])
    labels[s1] = K                                   # the second image has no target: no gt_boxes field
    gt[s1] = prop[s1]
    props = []
    for i, (b, c, t, ww) in enumerate(zip(prop.split(counts), labels.split(counts), gt.split(counts), w.split(counts))):
        inst = Instances((256, 256), proposal_boxes=Boxes(b.to(cuda)), gt_classes=c.to(cuda), gt_weights=ww.to(cuda))
        if i != 1:
            inst.gt_boxes = Boxes(t.to(cuda))
        props.append(inst)
    wc, bc = head.cls_score.weight.detach().cpu().clone().requires_grad_(), head.cls_score.bias.detach().cpu().clone().requires_grad_()
    wb, bb = head.bbox_pred.weight.detach().cpu().clone().requires_grad_(), head.bbox_pred.bias.detach().cpu().clone().requires_grad_()
    z0 = F.linear(x, wc, bc)
    d0 = F.linear(x, wb, bb) if reg else torch.zeros(R, 4 * K)
    lc0, lb0 = OM.oicr_losses(z0, d0, prop, dict(classes=labels, boxes=gt, weights=w))
    (lc0 + (lb0 if reg else 0)).backward()
    pred = head(x.to(cuda))
    assert pred[0].shape == (R, K + 1) and pred[1].shape == (R, 4 * K)
    if not reg:
        assert float(pred[1].abs().sum()) == 0.0
    out = head.losses(pred, props)
    assert set(out) == ({"loss_cls_r1", "loss_box_reg_r1"} if reg else {"loss_cls_r1"})
    rel_close(out["loss_cls_r1"], lc0, what="loss_cls")
    if reg:
        rel_close(out["loss_box_reg_r1"], lb0, what="loss_box_reg")
    sum(out.values()).backward()
    rel_close(head.cls_score.weight.grad, wc.grad, what="d cls_score.weight")
    rel_close(head.cls_score.bias.grad, bc.grad, what="d cls_score.bias")
    if reg:
        rel_close(head.bbox_pred.weight.grad, wb.grad, what="d bbox_pred.weight")
    fused = head.losses(pred, prop.to(cuda), labels.to(cuda), gt.to(cuda), w.to(cuda))
    for k in out:
        assert float(fused[k]) == float(out[k]), k
    empty = head.losses((pred[0][:0], pred[1][:0]), [])
    assert all(float(v) == 0.0 for v in empty.values())


def test_fused_mining_and_labelling_vs_oracle(cuda):
    """mine_top1 + match_label (jtsm_amd/csrc/mining.hip) against oracle/model.py's per-image torch
    restatement of get_pgt_top_k / label_and_sample_proposals: winning rows, labels and matched indices
    bit-exact; decoded boxes to fp32 rounding.  Ragged bags, an image without any present class, both the
    raw-proposal path (round 0) and the softmax + apply_deltas path (rounds k > 0)."""
    from jtsm_amd.layers.mining import match_label, mine_top1, pad_class_lists, row_lse

    g = torch.Generator().manual_seed(12)
    counts = [300, 41, 7]
    R, K = sum(counts), OM.NUM_THINGS
    boxes = torch.rand(R, 4, generator=g) * 300
    boxes[:, 2:] = boxes[:, :2] + 8 + torch.rand(R, 2, generator=g) * 150
    things = [torch.tensor([3, 17, 60]), torch.tensor([5]), torch.zeros(0, dtype=torch.int64)]
    probs = torch.rand(3, OM.NUM_MIL, generator=g)
    off = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32, device=cuda)
    cls, cnt, _ = pad_class_lists([t.to(cuda) for t in things], cuda)

    # round 0: scores given directly, boxes = proposals
    sc = torch.rand(R, OM.NUM_MIL, generator=g)
    pg = mine_top1(sc.to(cuda), boxes.to(cuda), off, cls, cnt, probs.to(cuda))
    lab = match_label(boxes.to(cuda), off, pg, cls, cnt, K)
    r0 = 0
    for i, n in enumerate(counts):
        b_i, s_i = boxes[r0:r0 + n], sc[r0:r0 + n]
        if things[i].numel():
            t = OM.mine_top1(b_i[:, None, :].expand(n, OM.NUM_MIL, 4), s_i, things[i], probs[i])
            assert torch.equal(pg["idx"][i, :things[i].numel()].cpu().long(), t["idx"])
            assert torch.equal(pg["boxes"][i, :things[i].numel()].cpu(), t["boxes"])
            assert torch.equal(pg["weights"][i, :things[i].numel()].cpu(), t["weights"])
            m = OM.match_and_label(b_i, t)
            assert torch.equal(lab["labels"][r0:r0 + n].cpu().long(), m["classes"])
            assert torch.equal(lab["matched"][r0:r0 + n].cpu().long(), m["idx"])
            assert torch.equal(lab["boxes"][r0:r0 + n].cpu(), m["boxes"])
            assert torch.equal(lab["weights"][r0:r0 + n].cpu(), m["weights"])
        else:
            assert (lab["labels"][r0:r0 + n] == K).all() and (lab["weights"][r0:r0 + n] == 0).all()
        r0 += n

    # rounds k > 0: softmax of logits + per-class decoded boxes, taken from a wider (fused) matrix
    wide = torch.randn(R, (K + 1) + 4 * K + 5, generator=g)
    z, d = wide[:, :K + 1], wide[:, K + 1:K + 1 + 4 * K] * 0.3
    wd = wide.to(cuda)
    zd, dd = wd[:, :K + 1], (wd[:, K + 1:K + 1 + 4 * K] * 0.3).contiguous()
    pg = mine_top1(zd, boxes.to(cuda), off, cls, cnt, probs.to(cuda), lse=row_lse(zd), deltas=dd)
    r0 = 0
    for i, n in enumerate(counts):
        if things[i].numel():
            ps = torch.softmax(z[r0:r0 + n], dim=-1)
            pb = OM.apply_deltas(d[r0:r0 + n], boxes[r0:r0 + n]).view(n, K, 4)
            t = OM.mine_top1(pb, ps, things[i], probs[i])
            assert torch.equal(pg["idx"][i, :things[i].numel()].cpu().long(), t["idx"])
            assert torch.allclose(pg["boxes"][i, :things[i].numel()].cpu(), t["boxes"], rtol=1e-5, atol=1e-3)
            assert torch.allclose(pg["scores"][i, :things[i].numel()].cpu(), t["scores"], rtol=1e-4, atol=1e-7)
        r0 += n


def test_fused_sgd_matches_torch_sgd(cuda):
    """jtsm_amd.solver.SGD (one launch for all parameters) against torch.optim.SGD: three steps with momentum,
    weight decay, two parameter groups, a channels_last 4-d weight and an odd-sized vector."""
    from jtsm_amd.solver import SGD

    g = torch.Generator().manual_seed(11)
    shapes = [(64, 32, 3, 3), (37,), (128, 64, 1, 1), (5, 7)]
    ref = [torch.randn(s, generator=g) for s in shapes]
    mine = [r.clone().to(cuda) for r in ref]
    mine[0] = mine[0].contiguous(memory_format=torch.channels_last)
    for t in ref + mine:
        t.requires_grad_(True)
    groups = lambda ps: [{"params": [ps[0], ps[2]], "lr": 0.05, "weight_decay": 5e-4},
                         {"params": [ps[1], ps[3]], "lr": 0.1, "weight_decay": 0.0}]
    o_ref = torch.optim.SGD(groups(ref), lr=0.05, momentum=0.9)
    o_mine = SGD(groups(mine), lr=0.05, momentum=0.9)
    for step in range(3):
        for r, m in zip(ref, mine):
            gr = torch.randn(r.shape, generator=g)
            r.grad = gr.clone()
            m.grad = gr.to(cuda)          # plain-contiguous gradient for the channels_last weight: copied to its layout
        v = mine[0]._version
        o_ref.step()
        o_mine.step()
        assert mine[0]._version > v      # the update is visible to version-keyed caches
        for r, m in zip(ref, mine):
            assert torch.allclose(m.detach().cpu(), r.detach(), rtol=1e-6, atol=1e-7), (step, tuple(r.shape))
    sd = o_mine.state_dict()
    assert len(sd["state"]) == 4 and "momentum_buffer" in next(iter(sd["state"].values()))


def test_fused_sgd_parameter_joining_late_matches_torch_sgd(cuda):
    """A parameter whose first gradient arrives at a LATER step than the others (a head whose loss was skipped, a
    layer unfrozen mid-training): torch.optim.SGD initialises momentum per parameter, and so does the fused update."""
    from jtsm_amd.solver import SGD

    g = torch.Generator().manual_seed(12)
    ref = [torch.randn(40, 8, generator=g), torch.randn(17, generator=g)]
    mine = [r.clone().to(cuda) for r in ref]
    for t in ref + mine:
        t.requires_grad_(True)
    o_ref = torch.optim.SGD(ref, lr=0.05, momentum=0.9, weight_decay=1e-3)
    o_mine = SGD(mine, lr=0.05, momentum=0.9, weight_decay=1e-3)
    for step in range(4):
        for i, (r, m) in enumerate(zip(ref, mine)):
            if i == 1 and step < 2:          # the second parameter gets no gradient during the first two steps
                r.grad = m.grad = None
                continue
            gr = torch.randn(r.shape, generator=g)
            r.grad, m.grad = gr.clone(), gr.to(cuda)
        o_ref.step()
        o_mine.step()
        for r, m in zip(ref, mine):
            assert torch.allclose(m.detach().cpu(), r.detach(), rtol=1e-6, atol=1e-7), step


@pytest.mark.parametrize("classes,pad", [(80, 0), (5, 3), (1, 0)])
def test_mask_bce_matches_torch(cuda, classes, pad):
    """mask_bce_loss (one launch each way) vs F.binary_cross_entropy_with_logits on the gathered class channel,
    incl. a padded channel pitch and the class-agnostic case."""
    from jtsm_amd.layers.wsl_losses import mask_bce_loss

    g = torch.Generator().manual_seed(21)
    n, side = 13, 28
    z = torch.randn(n, classes + pad, side, side, generator=g) * 3
    cls = torch.randint(0, classes, (n,), generator=g)
    t = torch.rand(n, side, side, generator=g) > 0.6
    z0 = z[:, :classes].clone().requires_grad_(True)
    sel = z0[torch.arange(n), cls] if classes > 1 else z0[:, 0]
    l0 = torch.nn.functional.binary_cross_entropy_with_logits(sel, t.float(), reduction="mean")
    (l0 * 1.7).backward()
    zd = z.to(cuda).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    l = mask_bce_loss(zd[:, :classes], cls.to(cuda), t.to(cuda))
    (l * 1.7).backward()
    assert abs(float(l) - float(l0)) <= 1e-6 * abs(float(l0))
    gd = zd.grad.cpu()
    assert torch.allclose(gd[:, :classes], z0.grad, rtol=1e-5, atol=1e-9)
    assert float(gd[:, classes:].abs().sum()) == 0.0


def test_paste_crop_targets_bit_exact_vs_paste_then_roi_align(cuda):
    """The mask refinery's targets (get_pgt_mask, roi_heads_jtsm.py:1997-2022): class probability pasted into the image
    at the box, cropped back at 28x28 by BitMasks.crop_and_resize.  The kernel evaluates the pasted image analytically
    at every ROIAlign tap (nothing is rasterised); the oracle rasterises the full-size paste (oracle/inference.py:
    paste_masks_in_image) and runs the C restatement of ROIAlign over it.  Bits must agree exactly — boxes inside,
    across and partly outside the image, tiny and large."""
    from jtsm_amd.layers.mining import paste_crop_targets
    from oracle import pooling as P
    from oracle.inference import paste_masks_in_image

    g = torch.Generator().manual_seed(21)
    H, W, n = 700, 820, 24
    prob = torch.rand(n, 28, 28, generator=g)
    prob[3] = 1.0
    prob[4] = 0.0
    cx, cy = torch.rand(n, generator=g) * W, torch.rand(n, generator=g) * H
    bw, bh = torch.rand(n, generator=g) * 260 + 3, torch.rand(n, generator=g) * 260 + 3
    boxes = torch.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], 1)
    boxes[0] = torch.tensor([-40.0, -30.0, 90.5, 120.25])          # partly outside
    boxes[1] = torch.tensor([5.0, 6.0, 7.5, 9.0])                  # tiny
    boxes[2] = torch.tensor([10.0, 12.0, 800.0, 690.0])            # nearly the whole image
    got = paste_crop_targets(prob.to(cuda), boxes.to(cuda), 28, H, W, 0.5).cpu()
    pasted = paste_masks_in_image(prob, boxes, (H, W), 0.5).to(torch.float32)
    rr = torch.cat([torch.arange(n, dtype=torch.float32)[:, None], boxes], 1)
    want = torch.from_numpy(P.roi_align_forward(pasted[:, None].numpy(), rr.numpy(), 1.0, 28, 28, 0, True))[:, 0] >= 0.5
    assert got.shape == want.shape and got.dtype == torch.bool
    assert torch.equal(got, want), int((got != want).sum())
    assert 0.2 < float(want.float().mean()) < 0.8
