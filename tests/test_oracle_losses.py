"""CPU suite: pins the torch-CPU loss restatements in oracle/model.py.

The reference holds one usable known answer on this arithmetic: tests/modeling/test_fast_rcnn.py:18-45
(loss_cls = 1.7951188087, loss_box_reg = 4.0357131958 at torch.manual_seed(132)) — CE over K+1 logits and
L1 (smooth-L1 with beta 0) on the gt-class deltas divided by R, with Box2BoxTransform (10,10,5,5).  The
OICR branch loss is the same arithmetic with per-row weights, so with weights = 1 it must reproduce
those numbers.  MIL (fast_rcnn_tsm.py) has no reference test: "parity unpinned", checked against an
independent loop formulation and its analytic gradient against autograd.
"""
import torch
import torch.nn.functional as F

from oracle import model as OM


def test_oicr_loss_reproduces_reference_fast_rcnn_pin():
    torch.manual_seed(132)
    cls_score = torch.nn.Linear(8, 6)      # construction order and RNG draws as in FastRCNNOutputLayers.__init__
    bbox_pred = torch.nn.Linear(8, 20)
    torch.nn.init.normal_(cls_score.weight, std=0.01)
    torch.nn.init.normal_(bbox_pred.weight, std=0.001)
    torch.nn.init.constant_(cls_score.bias, 0)
    torch.nn.init.constant_(bbox_pred.bias, 0)
    feat = torch.rand(2, 8)
    z, d = cls_score(feat), bbox_pred(feat)
    prop = torch.tensor([[0.8, 1.1, 3.2, 2.8], [2.3, 2.5, 7, 8]])
    gt = torch.tensor([[1, 1, 3, 3], [2, 2, 6, 6.0]])
    lab = dict(classes=torch.tensor([1, 2]), boxes=gt, weights=torch.ones(2))
    old = OM.NUM_THINGS
    OM.NUM_THINGS = 5
    try:
        lc, lb = OM.oicr_losses(z, d, prop, lab)
    finally:
        OM.NUM_THINGS = old
    assert torch.allclose(lc, torch.tensor(1.7951188087))
    assert torch.allclose(lb, torch.tensor(4.0357131958))


def test_apply_deltas_inverts_get_deltas():
    g = torch.Generator().manual_seed(0)
    src = torch.rand(50, 4, generator=g) * 50
    src[:, 2:] += src[:, :2] + 1
    tgt = torch.rand(50, 4, generator=g) * 50
    tgt[:, 2:] += tgt[:, :2] + 1
    assert torch.allclose(OM.apply_deltas(OM.box_deltas(src, tgt), src), tgt, atol=1e-3)


def _mil_loops(C, D, counts, y):
    """Literal per-image, per-class loops of SURVEY Appendix B (float64)."""
    C, D = C.double(), D.double()
    out, probs, r0 = torch.zeros_like(C), [], 0
    for n in counts:
        c, d = C[r0:r0 + n], D[r0:r0 + n]
        a = torch.exp(c - c.max(1, keepdim=True).values)
        a = a / a.sum(1, keepdim=True)
        b = torch.exp(d - d.max(0, keepdim=True).values)
        b = b / b.sum(0, keepdim=True)
        out[r0:r0 + n] = a * b
        probs.append((a * b).sum(0).clamp(1e-6, 1 - 1e-6))
        r0 += n
    p = torch.stack(probs)
    loss = -(y * p.log() + (1 - y) * (1 - p).log()).mean()
    return out, p, loss


def test_mil_restated_matches_loops_and_analytic_gradient():
    g = torch.Generator().manual_seed(1)
    counts = [37, 5, 120]
    R, nc = sum(counts), OM.NUM_MIL
    C = (torch.randn(R, nc, generator=g) * 2).requires_grad_()
    D = (torch.randn(R, nc, generator=g) * 2).requires_grad_()
    y = (torch.rand(3, nc, generator=g) < 0.1).float()
    s = OM.mil_scores(C, D, counts)
    p = OM.mil_image_probs(s, counts)
    loss = F.binary_cross_entropy(p, y)
    s2, p2, l2 = _mil_loops(C.detach(), D.detach(), counts, y.double())
    assert torch.allclose(s.double(), s2, atol=1e-7) and torch.allclose(p.double(), p2, atol=1e-6)
    assert abs(loss.item() - l2.item()) < 1e-5
    loss.backward()
    # analytic gradient of Appendix B
    Cd, Dd = C.detach().double(), D.detach().double()
    gC, gD, r0 = torch.zeros_like(Cd), torch.zeros_like(Dd), 0
    for i, n in enumerate(counts):
        a = F.softmax(Cd[r0:r0 + n], 1)
        b = F.softmax(Dd[r0:r0 + n], 0)
        sc = a * b
        ps = sc.sum(0)
        pc = ps.clamp(1e-6, 1 - 1e-6)
        gp = (pc - y[i].double()) / (pc * (1 - pc)) / (3 * nc) * ((ps >= 1e-6) & (ps <= 1 - 1e-6))
        gC[r0:r0 + n] = gp * sc - a * (gp * sc).sum(1, keepdim=True)
        gD[r0:r0 + n] = gp * (sc - b * ps)
        r0 += n
    assert torch.allclose(C.grad.double(), gC, atol=1e-7) and torch.allclose(D.grad.double(), gD, atol=1e-7)


def test_oracle_dc5_architecture_runs_and_trains_heads_only():
    """oracle/model.py's `dc5` architecture (the shipped single-level configuration, ResNet-WS v2 18): finite
    losses without a semantic term, stride-8 feature map, gradients for the heads."""
    import torch

    from oracle import model as OM

    p = OM.init_params_dc5(seed=1, depth=18, nt=20, ns=2, dan_dims=(64, 64), input_gain=1.0 / 64)
    x = OM.preprocess(p, [torch.rand(3, 64, 96) * 255], 8)
    f = OM.wsr_v2_dc5(p, x, 18)
    assert tuple(f.shape) == (1, 512, 8, 12)
    batch = OM.synthetic_batch(5, B=1, size=128, R=40, sp_block=8, n_stuff=1, nt=20, ns=2)
    names = [k for k in p if k.startswith("roi_heads.")]
    for n in names:
        p[n].requires_grad_(True)
    losses = OM.forward_losses(p, batch, depth=18, arch="dc5", nt=20, ns=2)
    assert "loss_sem_seg" not in losses and {"loss_cls", "loss_cls_r3", "loss_box_reg_r3", "loss_mask"} <= set(losses)
    total = sum(losses.values())
    assert bool(torch.isfinite(total))
    total.backward()
    assert all(p[n].grad is not None for n in names if "mask_refinery" not in n or float(losses["loss_mask"]) > 0)


def _mil_literal(C, D, counts, y, mean_loss=True):
    """A second, independently written reading of the MIL path — scalar Python floats (fp64), one line of the
    reference per step, no tensor operations:
      fast_rcnn_tsm.py:572-586   scores[r][c] = softmax over CLASSES of C[r] (dim=1)  x  softmax over the image's
                                 PROPOSALS of D[:, c] (dim=0), image by image
      :840-854 / :364-378        p_img[i][c] = clamp(sum_r scores, 1e-6, 1 - 1e-6)
      :346-362                   BCE(p_img, y): mean over all (image, class) entries, or sum / number of images
    Returns (scores, p_img, loss) as nested lists / float."""
    import math
    R, nc = len(C), len(C[0])
    scores = [[0.0] * nc for _ in range(R)]
    p_img, r0 = [], 0
    for n in counts:
        rows = range(r0, r0 + n)
        cls_soft = {}
        for r in rows:                                     # F.softmax(c, dim=1)
            m = max(C[r])
            e = [math.exp(v - m) for v in C[r]]
            z = sum(e)
            cls_soft[r] = [v / z for v in e]
        for c in range(nc):                                # F.softmax(d, dim=0)
            col = [D[r][c] for r in rows]
            if not col:
                continue
            m = max(col)
            e = [math.exp(v - m) for v in col]
            z = sum(e)
            for k, r in enumerate(rows):
                scores[r][c] = cls_soft[r][c] * (e[k] / z)
        p_img.append([min(max(sum(scores[r][c] for r in rows), 1e-6), 1.0 - 1e-6) for c in range(nc)])
        r0 += n
    total = 0.0
    for i in range(len(counts)):
        for c in range(nc):
            total += -(y[i][c] * math.log(p_img[i][c]) + (1.0 - y[i][c]) * math.log(1.0 - p_img[i][c]))
    loss = total / (len(counts) * nc) if mean_loss else total / len(counts)
    return scores, p_img, loss


def test_mil_second_literal_reading_agrees_incl_gradient():
    """The scalar reading above against oracle/model.py (values), and its central finite differences (fp64) against
    the oracle's autograd gradient: two independent readings of the unpinned MIL arithmetic that agree."""
    g = torch.Generator().manual_seed(3)
    counts = [7, 0, 12]                                   # ragged, incl. an empty bag
    R, nc = sum(counts), 9
    C = (torch.randn(R, nc, generator=g, dtype=torch.float64) * 2).requires_grad_()
    D = (torch.randn(R, nc, generator=g, dtype=torch.float64) * 2).requires_grad_()
    y = (torch.rand(len(counts), nc, generator=g) < 0.3).double()
    for mean_loss in (True, False):
        s = OM.mil_scores(C, D, counts)
        p = OM.mil_image_probs(s, counts)
        loss = F.binary_cross_entropy(p, y, reduction="mean" if mean_loss else "sum") / (1 if mean_loss else len(counts))
        s2, p2, l2 = _mil_literal(C.tolist(), D.tolist(), counts, y.tolist(), mean_loss)
        assert torch.allclose(s.detach(), torch.tensor(s2, dtype=torch.float64), atol=1e-12)
        assert torch.allclose(p.detach(), torch.tensor(p2, dtype=torch.float64), atol=1e-12)
        assert abs(float(loss) - l2) < 1e-12
        C.grad = D.grad = None
        loss.backward()
        eps = 1e-6
        for (name, T) in (("C", C), ("D", D)):
            for (r, c) in ((0, 0), (3, 4), (8, 8), (R - 1, 2)):
                base = T.detach().clone()
                vals = []
                for sgn in (+1, -1):
                    t = base.clone()
                    t[r, c] += sgn * eps
                    args = (t.tolist(), D.tolist()) if name == "C" else (C.tolist(), t.tolist())
                    vals.append(_mil_literal(args[0], args[1], counts, y.tolist(), mean_loss)[2])
                fd = (vals[0] - vals[1]) / (2 * eps)
                assert abs(fd - float(T.grad[r, c])) <= 1e-6 * max(1.0, abs(fd)), (name, r, c, fd, float(T.grad[r, c]))
