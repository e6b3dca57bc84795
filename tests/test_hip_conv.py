"""GPU suite: fp32-MFMA implicit-GEMM convolution / linear kernels vs the torch-CPU oracle
(oracle/nnref.py).  Bar (BASELINE.json): fp32 convs within 1e-4 relative.  The metric used is
max|a-b| / max|b| per tensor (<= 1e-4), plus an element-wise allclose with rtol 1e-4 and an atol
scaled to the tensor's magnitude.  Products are exact fp32 in both; only summation order differs.
"""
import numpy as np
import pytest
import torch

from oracle import nnref

pytestmark = pytest.mark.gpu

from jtsm_amd.layers import conv as K  # noqa: E402
from jtsm_amd.layers.elementwise import channel_sum, relu_backward  # noqa: E402

CL = torch.channels_last
REL = 1e-4
# fp16 operands round at 2^-11 = 4.9e-4 relative each; a contraction's error relative to its largest output stays
# below 2e-3 on these cases (measured ~3e-4): the stated fp16 tolerance of BASELINE configs[4]'s extra leg.
REL_BY_MATH = {"f32": 1e-4, "bf16x3": 1e-4, "f16": 2e-3}


@pytest.fixture(autouse=True, params=["f32", "bf16x3", "f16"])
def conv_math(request):
    """Every case runs in every contraction arithmetic: exact fp32 MFMA and split-bf16 (three bf16 MFMA
    products per fp32 product, csrc/conv_x3.h) at the same 1e-4 bar, and the fp16 path (one fp16 plane per operand,
    fp32 accumulate) at its own stated tolerance."""
    global REL
    old = K.MATH
    K.set_math(request.param)
    K.planes_clear()
    REL = REL_BY_MATH[request.param]
    yield request.param
    REL = 1e-4
    K.set_math(old)
    K.planes_clear()


def close(a, b, what="", chain=False):
    """chain: the value went through ReLU gates computed from forward activations.  In fp16 a pre-activation within
    2^-11 of zero can land on the other side of the gate than the fp32 oracle's, which changes a gradient term
    outright at that element (a discrete effect, not an arithmetic error): the fp16 bar for such tensors is a
    relative L2 error of 5e-2 instead of the max-norm bar."""
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if chain and K.MATH == "f16":
        l2 = float((a - b).norm() / (b.norm() + 1e-30))
        assert l2 <= 5e-2, "%s: relative L2 error %.3e" % (what, l2)
        return
    ref = b.abs().max().item() + 1e-30
    err = (a - b).abs().max().item()
    assert err <= REL * ref, "%s: max err %.3e vs max ref %.3e (rel %.2e)" % (what, err, ref, err / ref)


CASES = [
    # name,            N, C,  H,  W,  O,  k, s, p, d
    ("1x1",            2, 64, 20, 24, 256, 1, 1, 0, 1),
    ("1x1_s2",         2, 256, 21, 18, 128, 1, 2, 0, 1),
    ("3x3",            2, 64, 19, 23, 64, 3, 1, 1, 1),
    ("3x3_wide",       1, 128, 16, 16, 128, 3, 1, 1, 1),
    ("3x3_dil2",       1, 64, 17, 15, 128, 3, 1, 2, 2),
    ("3x3_s2",         1, 32, 17, 18, 96, 3, 2, 1, 1),
    ("stem7x7",        2, 4, 37, 41, 64, 7, 2, 3, 1),
    ("ntail80",        3, 256, 14, 14, 80, 1, 1, 0, 1),
    ("ntail54",        1, 128, 32, 32, 54, 1, 1, 0, 1),
    ("small_k",        1, 8, 9, 9, 12, 3, 1, 1, 1),
    # 64 x 64 tiles with the four-stage ring (csrc/conv_x3.h, NBUF = 4): whole rings, slices of 6 + 5 stages with a row
    # tail, and a 3x3 layer on a small map (tap-permuted K order; slices of 5, 5, 5 and 3 stages — shorter than the ring)
    ("ring_1x1",       2, 1024, 16, 16, 256, 1, 1, 0, 1),
    ("ring_tail",      1, 352, 20, 20, 128, 1, 1, 0, 1),
    ("ring_3x3",       1, 64, 12, 12, 128, 3, 1, 1, 1),
]


def make(case, seed=0):
    _, N, C_, H, W, O, k, s, p, d = case
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, C_, H, W, generator=g)
    w = torch.randn(O, C_, k, k, generator=g) * (2.0 / (C_ * k * k)) ** 0.5
    return x, w, s, p, d


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_forward_plain_and_fused(cuda, case):
    x, w, s, p, d = make(case)
    O = w.shape[0]
    g = torch.Generator().manual_seed(1)
    scale, bias = torch.rand(O, generator=g) + 0.5, torch.randn(O, generator=g)
    y0 = nnref.conv_bn_act(x, w, s, p, d)
    res = torch.randn(y0.shape, generator=g)
    xd, wd = x.to(cuda).contiguous(memory_format=CL), w.to(cuda).contiguous(memory_format=CL)
    y = K.conv2d_forward(xd, wd, s, p, d)
    assert y.is_contiguous(memory_format=CL)
    close(y, y0, "plain")
    y = K.conv2d_forward(xd, wd, s, p, d, scale.to(cuda), bias.to(cuda), res.to(cuda), True)
    close(y, nnref.conv_bn_act(x, w, s, p, d, scale, bias, res, True), "fused")


@pytest.mark.parametrize("case", [c for c in CASES if c[5] % 4 == 0], ids=[c[0] for c in CASES if c[5] % 4 == 0])
def test_conv_backward_data_and_weight(cuda, case):
    x, w, s, p, d = make(case)
    x.requires_grad_(True)
    w.requires_grad_(True)
    O = w.shape[0]
    g = torch.Generator().manual_seed(2)
    scale = torch.rand(O, generator=g) + 0.5
    y0 = nnref.conv_bn_act(x, w, s, p, d, scale)
    dy = torch.randn(y0.shape, generator=g)
    y0.backward(dy)
    dyd = dy.to(cuda).contiguous(memory_format=CL)
    xd, wd = x.detach().to(cuda).contiguous(memory_format=CL), w.detach().to(cuda).contiguous(memory_format=CL)
    dx = K.conv2d_backward_data(dyd, wd, tuple(x.shape), s, p, d, kscale=scale.to(cuda))
    close(dx, x.grad, "dgrad")
    dw = K.conv2d_backward_weight(dyd, xd, tuple(w.shape), s, p, d, row_scale=scale.to(cuda))
    assert dw.is_contiguous(memory_format=CL)
    close(dw, w.grad, "wgrad")
    # the bias gradient beside dW (in the plane arithmetics: inside the same contraction, x3_bias_mma) — every case
    # crosses a different kernel / slice plan: generic 128 and 256 tiles, the 3x3 halo kernel, one slice and many
    db = torch.full((O,), float("nan"), device=cuda)
    dw2 = K.conv2d_backward_weight(dyd, xd, tuple(w.shape), s, p, d, bias_out=db)
    close(db, dy.double().sum((0, 2, 3)), "bias gradient")
    close(dw2, w.grad / scale.view(-1, 1, 1, 1), "wgrad beside the bias gradient")
    # row scale applied to the data gradient's rows (jtsm_conv2d_backward_data_ex_*) for the 1x1 cases
    # accumulate + relu gate epilogue of dgrad
    acc = torch.randn(x.shape, generator=g)
    gate = torch.randn(x.shape, generator=g)
    dx2 = K.conv2d_backward_data(dyd, wd, tuple(x.shape), s, p, d, kscale=scale.to(cuda),
                                 accumulate=acc.to(cuda), relu_mask=gate.to(cuda))
    close(dx2, (x.grad + acc) * (gate > 0), "dgrad+acc+gate")


def test_autograd_bottleneck_like_chain(cuda):
    """conv1x1-bn-relu -> conv3x3-bn-relu -> conv1x1-bn (+shortcut) relu, all through
    conv2d_fused, gradients vs torch-CPU autograd of the same graph."""
    g = torch.Generator().manual_seed(3)
    N, C0, C1, H, W = 2, 64, 32, 14, 12
    x = torch.randn(N, C0, H, W, generator=g, requires_grad=True)
    ws = [torch.randn(C1, C0, 1, 1, generator=g) * 0.2, torch.randn(C1, C1, 3, 3, generator=g) * 0.1,
          torch.randn(C0, C1, 1, 1, generator=g) * 0.2]
    for t in ws:
        t.requires_grad_(True)
    sb = [(torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1) for c in (C1, C1, C0)]
    o = nnref.conv_bn_act(x, ws[0], 1, 0, 1, *sb[0], None, True)
    o = nnref.conv_bn_act(o, ws[1], 1, 1, 1, *sb[1], None, True)
    y0 = nnref.conv_bn_act(o, ws[2], 1, 0, 1, *sb[2], x, True)
    dy = torch.randn(y0.shape, generator=g)
    y0.backward(dy)

    xd = x.detach().to(cuda).contiguous(memory_format=CL).requires_grad_(True)
    wd = [t.detach().to(cuda).contiguous(memory_format=CL).requires_grad_(True) for t in ws]
    sbd = [(a.to(cuda), b.to(cuda)) for a, b in sb]
    o = K.conv2d_fused(xd, wd[0], *sbd[0], None, 1, 0, 1, True)
    o = K.conv2d_fused(o, wd[1], *sbd[1], None, 1, 1, 1, True)
    y = K.conv2d_fused(o, wd[2], *sbd[2], xd, 1, 0, 1, True)
    close(y, y0, "fwd")
    y.backward(dy.to(cuda))
    close(xd.grad, x.grad, "dx", chain=True)
    for i in range(3):
        close(wd[i].grad, ws[i].grad, "dw%d" % i, chain=True)


def test_linear_dan_shape_slice(cuda):
    """DAN fc1 geometry (K = 256*7*7 = 12544 -> 2048) on a slice of rows, + bias + relu + grads."""
    g = torch.Generator().manual_seed(4)
    R, Kd, O = 192, 12544, 256
    x = torch.randn(R, Kd, generator=g, requires_grad=True)
    w = (torch.randn(O, Kd, generator=g) * 0.005).requires_grad_(True)
    b = torch.full((O,), 0.1, requires_grad=True)
    y0 = nnref.linear(x, w, b, True)
    dy = torch.randn(y0.shape, generator=g)
    # an output within rounding of the ReLU kink may be gated differently by two correct implementations:
    # send no gradient through such elements, so the comparison is about arithmetic, not about ties
    dy[(y0.detach().abs() < 1e-3) & (y0.detach() != 0)] = 0
    dy[(y0.detach() == 0) & ((x.detach() @ w.detach().t() + b.detach()).abs() < 1e-3)] = 0
    y0.backward(dy)
    xd = x.detach().to(cuda).requires_grad_(True)
    wd = w.detach().to(cuda).requires_grad_(True)
    bd = b.detach().to(cuda).requires_grad_(True)
    y = K.linear_fused(xd, wd, bd, True)
    close(y, y0, "fc fwd")
    y.backward(dy.to(cuda))
    close(xd.grad, x.grad, "fc dx")
    close(wd.grad, w.grad, "fc dw")
    close(bd.grad, b.grad, "fc db")


def test_backbone_layer_shapes_vs_torch_gpu(cuda):
    """Full-size layers of R50-FPN at 2 x 1024^2 (res3 3x3, res4 1x1, FPN output 3x3 at p3) against
    torch's own GPU convolution on the same NHWC tensors (fp32)."""
    gen = torch.Generator(device=cuda).manual_seed(5)
    for (C_, O, HW, k, s, p) in [(128, 128, 128, 3, 1, 1), (1024, 256, 64, 1, 1, 0), (256, 256, 128, 3, 1, 1),
                                  (512, 1024, 128, 1, 2, 0)]:
        x = torch.randn(2, C_, HW, HW, device=cuda, generator=gen).contiguous(memory_format=CL)
        w = (torch.randn(O, C_, k, k, device=cuda, generator=gen) * (2.0 / (C_ * k * k)) ** 0.5).contiguous(memory_format=CL)
        y = K.conv2d_forward(x, w, s, p, 1)
        y0 = torch.nn.functional.conv2d(x.double(), w.double(), None, s, p)
        close(y, y0, "fwd %s" % ((C_, O, HW, k),))
        dy = torch.randn(y.shape, device=cuda, generator=gen).contiguous(memory_format=CL)
        dx0, dw0 = torch.autograd.grad(
            torch.nn.functional.conv2d(x.double().requires_grad_(), w.double().requires_grad_(), None, s, p),
            [], dy.double(), allow_unused=True) if False else (None, None)
        xr, wr = x.double().requires_grad_(), w.double().requires_grad_()
        torch.nn.functional.conv2d(xr, wr, None, s, p).backward(dy.double())
        close(K.conv2d_backward_data(dy, w, tuple(x.shape), s, p, 1), xr.grad, "dgrad")
        close(K.conv2d_backward_weight(dy, x, tuple(w.shape), s, p, 1), wr.grad, "wgrad")


def test_relu_backward_and_channel_sum(cuda):
    g = torch.Generator().manual_seed(6)
    y = torch.randn(3, 20, 7, 9, generator=g).to(cuda).contiguous(memory_format=CL)
    dy = torch.randn(3, 20, 7, 9, generator=g).to(cuda).contiguous(memory_format=CL)
    assert torch.equal(relu_backward(dy, y), dy * (y > 0))
    close(channel_sum(dy), dy.sum((0, 2, 3)), "channel_sum")
    m = torch.randn(1000, 133, generator=g).to(cuda)
    close(channel_sum(m), m.sum(0), "channel_sum 2d")
    # every width / height class of the slab scheme (narrow, not a multiple of 4, wider than a workgroup, few rows,
    # the float4 form), each reproducible bit for bit: bias gradients carry no atomics
    for rows, ch in ((5000, 56), (70000, 80), (37, 256), (300, 1027), (4096, 256), (1, 8), (129, 2048)):
        m = torch.randn(rows, ch, generator=g).to(cuda)
        a = channel_sum(m)
        close(a, m.double().sum(0), "channel_sum %dx%d" % (rows, ch))
        assert torch.equal(a, channel_sum(m)) and torch.equal(a, channel_sum(m.clone()))


def test_spatial_helpers_match_torch(cuda):
    """max_pool 3x3/s2/p1, nearest-x2 + add, stride-2 subsample: bit-exact vs torch's own ops
    (same arithmetic, no rounding freedom) incl. backward."""
    import torch.nn.functional as F

    from jtsm_amd.layers.elementwise import max_pool_3x3_s2, subsample2, upsample2_add

    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 8, 13, 18, generator=g)
    xd = x.to(cuda).contiguous(memory_format=CL).requires_grad_()
    xr = x.clone().requires_grad_()
    y, y0 = max_pool_3x3_s2(xd), F.max_pool2d(xr, 3, 2, 1)
    assert torch.equal(y.cpu(), y0)
    gy = torch.randn(y0.shape, generator=g)
    y.backward(gy.to(cuda)); y0.backward(gy)
    assert torch.allclose(xd.grad.cpu(), xr.grad, atol=1e-6)

    top = torch.randn(2, 8, 5, 7, generator=g)
    lat = torch.randn(2, 8, 10, 14, generator=g)
    td, ld_ = top.to(cuda).requires_grad_(), lat.to(cuda).requires_grad_()
    tr, lr = top.clone().requires_grad_(), lat.clone().requires_grad_()
    o, o0 = upsample2_add(td, ld_), lr + F.interpolate(tr, scale_factor=2.0, mode="nearest")
    assert torch.equal(o.cpu(), o0)
    go = torch.randn(o0.shape, generator=g)
    o.backward(go.to(cuda)); o0.backward(go)
    assert torch.allclose(td.grad.cpu(), tr.grad, atol=1e-6) and torch.equal(ld_.grad.cpu(), lr.grad)

    for hw in ((32, 32), (7, 9)):
        x = torch.randn(1, 4, *hw, generator=g)
        xd, xr = x.to(cuda).requires_grad_(), x.clone().requires_grad_()
        s, s0 = subsample2(xd), F.max_pool2d(xr, 1, 2, 0)
        assert torch.equal(s.cpu(), s0)
        gs = torch.randn(s0.shape, generator=g)
        s.backward(gs.to(cuda)); s0.backward(gs)
        assert torch.equal(xd.grad.cpu(), xr.grad)


def test_group_norm_relu_and_bilinear2x_vs_torch_cpu(cuda):
    """SemSegFPNHead's non-GEMM layers: fused GroupNorm(32)+ReLU and bilinear x2 (align_corners=False),
    forward and backward, against the stock torch ops the reference uses (CPU)."""
    import torch.nn.functional as F

    from jtsm_amd.layers.elementwise import group_norm_relu, upsample_bilinear2x

    g = torch.Generator().manual_seed(10)
    for (n, c, h, w, relu) in [(2, 128, 33, 40, True), (1, 128, 64, 64, False), (2, 256, 9, 7, True)]:
        x = torch.randn(n, c, h, w, generator=g) * 2 + 0.5
        ga, be = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
        xr, gar, ber = x.clone().requires_grad_(), ga.clone().requires_grad_(), be.clone().requires_grad_()
        y0 = F.group_norm(xr, 32, gar, ber, 1e-5)
        y0 = F.relu(y0) if relu else y0
        dy = torch.randn(y0.shape, generator=g)
        y0.backward(dy)
        xd = x.to(cuda).contiguous(memory_format=CL).requires_grad_()
        gad, bed = ga.to(cuda).requires_grad_(), be.to(cuda).requires_grad_()
        y = group_norm_relu(xd, gad, bed, 32, 1e-5, relu)
        close(y, y0, "gn fwd")
        y.backward(dy.to(cuda))
        close(xd.grad, xr.grad, "gn dx")
        close(gad.grad, gar.grad, "gn dgamma")
        close(bed.grad, ber.grad, "gn dbeta")
    for (n, c, h, w) in [(2, 128, 16, 20), (1, 8, 1, 5), (1, 4, 7, 1)]:
        x = torch.randn(n, c, h, w, generator=g)
        xr = x.clone().requires_grad_()
        y0 = F.interpolate(xr, scale_factor=2, mode="bilinear", align_corners=False)
        dy = torch.randn(y0.shape, generator=g)
        y0.backward(dy)
        xd = x.to(cuda).contiguous(memory_format=CL).requires_grad_()
        y = upsample_bilinear2x(xd)
        close(y, y0, "up2 fwd")
        y.backward(dy.to(cuda))
        close(xd.grad, xr.grad, "up2 bwd")


def test_fused_upsample_cross_entropy_vs_torch_cpu(cuda):
    """semantic_seg.py:179-188: CE(interpolate(logits, x4, bilinear), target, ignore 255), fused."""
    import torch.nn.functional as F

    from jtsm_amd.layers.elementwise import semseg_cross_entropy

    g = torch.Generator().manual_seed(11)
    # (x4 runs the LDS-tiled backward: whole tiles, ragged tiles on both edges, maps smaller than one tile)
    for (n, c, hs, ws, scale) in [(2, 54, 16, 24, 4), (1, 7, 5, 3, 2), (1, 54, 8, 8, 4), (2, 54, 5, 7, 4),
                                  (1, 54, 3, 1, 4), (1, 13, 9, 2, 4), (1, 54, 1, 1, 4)]:
        z = torch.randn(n, c, hs, ws, generator=g) * 3
        t = torch.randint(0, c, (n, hs * scale, ws * scale), generator=g)
        t[:, :2] = 255
        t[0, 5:, 3] = 255
        zr = z.clone().requires_grad_()
        l0 = F.cross_entropy(F.interpolate(zr, scale_factor=scale, mode="bilinear", align_corners=False), t,
                             reduction="mean", ignore_index=255)
        (l0 * 0.7).backward()
        # a [:, :c] slice of a wider channels-last map, like the padded predictor output
        wide = torch.zeros(n, c + 2, hs, ws).contiguous(memory_format=CL)
        wide[:, :c] = z
        wd = wide.to(cuda).requires_grad_()
        loss = semseg_cross_entropy(wd[:, :c], t.to(cuda), scale, 255)
        close(loss, l0, "ce loss")
        (loss * 0.7).backward()
        close(wd.grad[:, :c], zr.grad, "ce dlogits")
        assert wd.grad[:, c:].abs().max().item() == 0


def _plane_values(pt):
    """fp32 values a PlaneTensor stands for (gradient planes: times 2^GRAD_SHIFT in fp16 mode — left in, both sides alike)."""
    buf = pt.buf
    if K.MATH == "f16":
        return buf[:pt.numel].view(torch.float16).float()
    half = buf.numel() // 2
    return buf[:pt.numel].view(torch.bfloat16).float() + buf[half:half + pt.numel].view(torch.bfloat16).float()


def test_bias_gradient_from_the_data_gradient_epilogue(cuda):
    """A layer's bias gradient = the column sums of its output gradient, taken by the launch that WRITES that gradient
    (planes_backward_data(bias_out=...)): the same planes as the plain call, the sums those of the planes' pass."""
    if K.MATH == "f32":
        pytest.skip("plane arithmetics only")
    g = torch.Generator().manual_seed(21)
    for (n, c, h, w, o, k, pad) in ((37, 256, 14, 14, 256, 3, 1), (9, 256, 28, 28, 80, 1, 0), (5, 64, 12, 20, 96, 3, 1)):
        dy = K.PlaneTensor.of((torch.randn(n, o, h, w, generator=g) * 0.1).to(cuda).contiguous(memory_format=CL), grad=True)
        wt = (torch.randn(o, c, k, k, generator=g) * 0.05).to(cuda).contiguous(memory_format=CL)
        gate = K.PlaneTensor.of(torch.randn(n, c, h, w, generator=g).to(cuda).contiguous(memory_format=CL))
        plain = K.planes_backward_data(dy, wt, (n, c, h, w), 1, pad, 1, gate=gate)
        batch = K.ColsumBatch()
        got, summed = K.planes_backward_data(dy, wt, (n, c, h, w), 1, pad, 1, gate=gate, bias_out=batch.slot("b"))
        # (the launch that sums runs one K slice on the generic tiling, the plain call may split K: same values to rounding)
        assert float((_plane_values(got) - _plane_values(plain)).abs().max()) <= (2e-3 if K.MATH == "f16" else 3e-5) * float(_plane_values(plain).abs().max())
        if not summed:
            assert batch.finish() == {}
            continue
        db = batch.finish()["b"]
        ref = K.planes_channel_sum(plain)
        # (ref sums the 16-bit planes, db the fp32 values they were rounded from: fp16 planes round at 2^-11)
        bar = 2e-3 if K.MATH == "f16" else 2e-4
        assert float((db - ref).abs().max()) <= bar * float(ref.abs().max()) + 1e-6, (n, c, h, w)
        batch2 = K.ColsumBatch()
        K.planes_backward_data(dy, wt, (n, c, h, w), 1, pad, 1, gate=gate, bias_out=batch2.slot("b"))
        assert torch.equal(batch2.finish()["b"], db)                         # reproducible bit for bit
    # the fold alone: ragged row counts, several widths
    import ctypes as C

    from jtsm_amd import _lib as L
    for width in (4, 80, 256, 1024):
        parts = [torch.randn(r, width, generator=g).to(cuda) for r in (1, 7, 965, 130, 2)]
        out = torch.empty(len(parts), width, device=cuda)
        ptrs = (C.c_void_p * len(parts))(*[p.data_ptr() for p in parts])
        rows = (C.c_int * len(parts))(*[p.shape[0] for p in parts])
        L.check(L.lib().jtsm_colsum_fold_f32(ptrs, rows, len(parts), width, L.ptr(out), L.stream()), "colsum_fold")
        for p, o_ in zip(parts, out):
            assert torch.allclose(o_.cpu(), p.cpu().double().sum(0).float(), rtol=1e-5, atol=1e-4)


def test_padded_predictor_hands_the_cross_entropy_gradient_through(cuda):
    """54 classes on 56-wide rows (semantic_seg.py:179-188 behind a 1x1 predictor): the slice behind the channel
    padding takes the loss's zero-padded gradient buffer as it is — same gradients as the torch ops, no zeros + copy."""
    import torch.nn.functional as F

    from jtsm_amd.layers import conv as K
    from jtsm_amd.layers.elementwise import semseg_cross_entropy

    g = torch.Generator().manual_seed(13)
    x = torch.randn(2, 128, 24, 40, generator=g)
    w = torch.randn(54, 128, 1, 1, generator=g) * 0.05
    b = torch.randn(54, generator=g) * 0.1
    t = torch.randint(0, 54, (2, 96, 160), generator=g)
    t[:, :3] = 255
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    ref = F.cross_entropy(F.interpolate(F.conv2d(xr, wr, br), scale_factor=4, mode="bilinear", align_corners=False), t,
                          ignore_index=255)
    ref.backward()
    xd = x.to(cuda).contiguous(memory_format=CL).requires_grad_()
    wd = w.to(cuda).contiguous(memory_format=CL).requires_grad_()
    bd = b.to(cuda).requires_grad_()
    before = K._LeadingChannels.passed_through
    y = K.conv2d_fused(xd, wd, None, bd, None, 1, 0, 1, False, True)
    assert tuple(y.shape) == (2, 54, 24, 40)
    loss = semseg_cross_entropy(y, t.to(cuda), 4, 255)
    loss.backward()
    assert K._LeadingChannels.passed_through == before + 1
    close(loss, ref, "loss")
    close(xd.grad, xr.grad, "dx")
    close(wd.grad, wr.grad, "dw")
    close(bd.grad, br.grad, "db")
    # a consumer that knows nothing of the padding still gets the ordinary (padded) gradient
    y2 = K.conv2d_fused(xd, wd, None, bd, None, 1, 0, 1, False, True)
    (y2 * 2.0).sum().backward()
    assert K._LeadingChannels.passed_through == before + 1


def test_tiled_cross_entropy_backward_is_the_plain_gather_bit_for_bit(cuda, monkeypatch):
    """The x4 backward through LDS (one soft-max per output pixel) adds the same products in the same order as the
    plain gather it replaces: equal bits, at the JTSM size and on ragged maps."""
    from jtsm_amd.layers.elementwise import semseg_cross_entropy

    g = torch.Generator().manual_seed(12)
    for (n, c, hs, ws) in [(2, 54, 256, 256), (1, 54, 37, 19), (1, 54, 2, 3)]:
        z = (torch.randn(n, c + 2, hs, ws, generator=g) * 3).to(cuda).contiguous(memory_format=CL)
        t = torch.randint(0, c, (n, hs * 4, ws * 4), generator=g)
        t[:, : hs] = 255
        t[0, :, 3::7] = 255
        t = t.to(cuda)
        grads = []
        for form in ("1", "0"):
            monkeypatch.setenv("JTSM_CE_BWD_TILED", form)
            zd = z.clone().requires_grad_()
            semseg_cross_entropy(zd[:, :c], t, 4, 255).backward()
            grads.append(zd.grad.clone())
        assert torch.equal(grads[0], grads[1]), (n, c, hs, ws, (grads[0] - grads[1]).abs().max().item())
        assert grads[0].abs().max().item() > 0


def test_empty_batch_through_conv_linear_and_mask_head(cuda):
    """Zero rows (an image batch with no foreground roi) must flow through forward AND backward and leave
    zero (not missing) gradients — DDP without find_unused_parameters relies on that."""
    from jtsm_amd.layers.shape_spec import ShapeSpec
    from jtsm_amd.modeling.roi_heads.mask_head import MaskRCNNConvUpsampleWSLHead, mask_rcnn_loss

    head = MaskRCNNConvUpsampleWSLHead(ShapeSpec(channels=16, height=14, width=14), num_classes=5,
                                       conv_dims=[16, 16, 16]).to(cuda)
    x = torch.zeros(0, 16, 14, 14, device=cuda).contiguous(memory_format=CL)
    logits, _ = head.layers(x)
    assert logits.shape == (0, 5, 28, 28)
    loss = mask_rcnn_loss(logits, torch.zeros(0, dtype=torch.int64, device=cuda),
                          torch.zeros(0, 28, 28, dtype=torch.bool, device=cuda))
    loss.backward()
    for n, p in head.named_parameters():
        assert p.grad is not None and float(p.grad.abs().sum()) == 0.0, n
    w = torch.randn(8, 12, device=cuda, requires_grad=True)
    y = K.linear_fused(torch.zeros(0, 12, device=cuda), w, None, True, False)
    y.sum().backward()
    assert y.shape == (0, 8) and float(w.grad.abs().sum()) == 0.0


@pytest.mark.parametrize("relu", [False, True])
def test_conv_transpose_2x2_matches_torch(cuda, relu):
    """ConvTranspose2d(kernel 2, stride 2) (+ ReLU) — the mask heads' upsampler — against torch's conv_transpose2d in
    fp64 on the CPU: output, input gradient, weight gradient (in the parameter's own channels_last strides) and bias
    gradient.  In the plane arithmetics this is the GEMM with the pixel-shuffle epilogue and the 2x2/stride-2
    convolution roles (layers/conv.py: conv_transpose2x2_*); in exact fp32 the GEMM + shuffle copy."""
    from jtsm_amd.layers.wrappers import ConvTranspose2d

    g = torch.Generator().manual_seed(5)
    n, ci, co, h, w = 7, 64, 96, 14, 14
    layer = ConvTranspose2d(ci, co, kernel_size=2, stride=2, padding=0)
    with torch.no_grad():
        layer.weight.copy_(torch.randn(ci, co, 2, 2, generator=g) * 0.1)
        layer.bias.copy_(torch.randn(co, generator=g) * 0.3)
    x0 = torch.randn(n, ci, h, w, generator=g)
    dy0 = torch.randn(n, co, 2 * h, 2 * w, generator=g)
    xr = x0.double().requires_grad_()
    wr, br = layer.weight.detach().double().requires_grad_(), layer.bias.detach().double().requires_grad_()
    yr = torch.nn.functional.conv_transpose2d(xr, wr, br, stride=2)
    if relu:
        yr = torch.relu(yr)
    yr.backward(dy0.double())

    layer = layer.to(cuda)
    assert layer.weight.permute(0, 2, 3, 1).is_contiguous()      # (in, kh, kw, out) storage
    x = x0.to(cuda).contiguous(memory_format=CL).requires_grad_()
    y = layer(x, relu=relu)
    assert y.shape == (n, co, 2 * h, 2 * w) and y.permute(0, 2, 3, 1).is_contiguous()
    y.backward(dy0.to(cuda).contiguous(memory_format=CL))
    if K.MATH != "f32":
        assert type(y.grad_fn).__name__ == "_ConvTranspose2x2Backward"
    close(y, yr, "conv_transpose y", chain=relu)
    close(x.grad, xr.grad, "conv_transpose dx", chain=relu)
    close(layer.weight.grad, wr.grad, "conv_transpose dw", chain=relu)
    close(layer.bias.grad, br.grad, "conv_transpose db", chain=relu)
    assert layer.weight.grad.stride() == layer.weight.stride()


def test_mask_tower_as_one_node_matches_layer_by_layer(cuda):
    """The mask head's layers as one autograd node (ReLU gates in the data-gradient epilogues, planes handed from
    launch to launch) against the same head run layer by layer: logits and every gradient.  Same kernels and the same
    summation orders, so the bar is the arithmetic's own (1e-4 relative; fp16: relative L2, gates may flip)."""
    from jtsm_amd.layers import fused_blocks
    from jtsm_amd.layers.shape_spec import ShapeSpec
    from jtsm_amd.modeling.roi_heads.mask_head import MaskRCNNConvUpsampleWSLHead

    torch.manual_seed(3)
    head = MaskRCNNConvUpsampleWSLHead(ShapeSpec(channels=64, height=14, width=14), num_classes=8,
                                       conv_dims=[64, 64, 64]).to(cuda)
    with torch.no_grad():
        for p in head.parameters():
            if p.dim() == 1:
                p.normal_(0, 0.2)
        head.predictor.weight.normal_(0, 0.1)
    x0 = torch.randn(9, 64, 14, 14, device=cuda).contiguous(memory_format=CL)
    dl = torch.randn(9, 8, 28, 28, device=cuda).contiguous(memory_format=CL)
    dup = torch.randn(9, 64, 28, 28, device=cuda).contiguous(memory_format=CL)
    out = {}
    for fused in (False, True):
        fused_blocks.ENABLED = fused
        try:
            K.planes_clear()
            head.zero_grad(set_to_none=True)
            x = x0.clone().requires_grad_()
            logits, up = head.layers(x)
            if K.MATH != "f32":
                assert (type(logits.grad_fn).__name__ == "_MaskTowerFnBackward") == fused
            logits.backward(dl)
            out[fused] = [logits.detach(), up.detach(), x.grad] + [p.grad.clone() for p in head.parameters()]
        finally:
            fused_blocks.ENABLED = True
    names = ["logits", "upsampled", "dx"] + [n for n, _ in head.named_parameters()]
    for n, a, b in zip(names, out[True], out[False]):
        close(a, b, "mask tower " + n, chain=True)
    if K.MATH != "f32":
        # gradients through BOTH outputs, and through the features alone (the logits unused)
        for use_logits in (True, False):
            res = {}
            for fused in (False, True):
                fused_blocks.ENABLED = fused
                try:
                    K.planes_clear()
                    head.zero_grad(set_to_none=True)
                    x = x0.clone().requires_grad_()
                    logits, up = head.layers(x)
                    loss = (up * dup).sum() + ((logits * dl).sum() if use_logits else 0.0)
                    loss.backward()
                    res[fused] = [x.grad] + [p.grad.clone() if p.grad is not None else None for p in head.parameters()]
                finally:
                    fused_blocks.ENABLED = True
            for n, a, b in zip(names[2:], res[True], res[False]):
                assert (a is None) == (b is None), n
                if a is not None:
                    close(a, b, "mask tower (features%s) %s" % (" + logits" if use_logits else " only", n), chain=True)
        # a caller that uses the logits only: the node writes no fp32 upsampled features at all
        head.return_features = False
        K.planes_clear()
        head.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_()
        logits, up = head.layers(x)
        assert up is None
        logits.backward(dl)
        got = [logits.detach(), None, x.grad] + [p.grad.clone() for p in head.parameters()]
        for n, a, b in zip(names, got, out[True]):
            if a is not None:
                assert torch.equal(a, b), n          # the same launches, minus one store


def test_fc_stack_as_one_node_matches_layer_by_layer(cuda):
    """The box head's FC stack as one autograd node — the per-roi rescale folded into the plane split and into fc1's
    data-gradient epilogue, ReLU + dropout gates as one pass — against the layer-by-layer form (explicit multiply,
    Linear + ReLU per layer).  Dropout off: same arithmetic, same summation orders, the arithmetic's own bar.  Then
    dropout on: the counter-based mask keeps ~(1 - p) of the units, scales the rest by 1 / (1 - p), is a function of
    the seed alone, and the backward gates with exactly that mask."""
    from jtsm_amd.layers import fused_blocks
    from jtsm_amd.layers.elementwise import dropout_split_
    from jtsm_amd.layers.shape_spec import ShapeSpec
    from jtsm_amd.modeling.roi_heads.box_head import DiscriminativeAdaptionNeck

    torch.manual_seed(4)
    head = DiscriminativeAdaptionNeck(ShapeSpec(channels=32, height=2, width=2), fc_dims=[64, 96]).to(cuda)
    with torch.no_grad():
        for fc in head.fcs:
            fc.weight.normal_(0, 0.1)
            fc.bias.normal_(0, 0.2)
    head.train()
    head.dropout_p = 0.0
    x0 = torch.randn(40, 32, 2, 2, device=cuda).contiguous(memory_format=CL)
    rs = torch.rand(40, device=cuda) + 0.5
    dout = torch.randn(40, 96, device=cuda)
    out = {}
    for fused in (False, True):
        fused_blocks.ENABLED = fused
        try:
            K.planes_clear()
            head.zero_grad(set_to_none=True)
            x = x0.clone().requires_grad_()
            y = head(x, roi_scale=rs)
            if K.MATH != "f32":
                assert (type(y.grad_fn).__name__ == "_FcStackFnBackward") == fused
            y.backward(dout)
            out[fused] = [y.detach(), x.grad] + [p.grad.clone() for p in head.parameters()]
        finally:
            fused_blocks.ENABLED = True
    names = ["out", "dx"] + [n for n, _ in head.named_parameters()]
    for n, a, b in zip(names, out[True], out[False]):
        close(a, b, "fc stack " + n, chain=True)
    # ---- the dropout pass on its own
    t = torch.rand(1 << 16, device=cuda) + 0.1
    a, b, c = t.clone(), t.clone(), t.clone()
    dropout_split_(a, 0.5, 1234)
    dropout_split_(b, 0.5, 1234)
    dropout_split_(c, 0.5, 1235)
    assert torch.equal(a, b) and not torch.equal(a, c)
    kept = a != 0
    assert abs(float(kept.float().mean()) - 0.5) < 0.02 and torch.equal(a[kept], (t * 2.0)[kept])
    if K.MATH == "f32":
        return
    # ---- dropout on, through the node: forward zeros = dropped or inactive units, gradient flows only through the rest
    head.dropout_p = 0.5
    K.planes_clear()
    head.zero_grad(set_to_none=True)
    x = x0.clone().requires_grad_()
    y = head(x, roi_scale=rs)
    y.backward(torch.ones_like(y))
    frac = float((y == 0).float().mean())
    assert 0.5 < frac < 0.95 and bool(torch.isfinite(x.grad).all())
    g_b2 = head.fcs[-1].bias.grad                 # = sum over rows of (y > 0) * 1 / (1 - p)
    close(g_b2, (y > 0).float().sum(0) * 2.0, "fc stack dropout bias gradient")


@pytest.mark.parametrize("shape", [
    (2, 256, 256, 256, 256, 3, 1, 1),      # FPN p2 output conv (256x256 tiles, no split)
    (2, 1024, 64, 64, 256, 1, 1, 0),       # res4 1x1 (split-K, wide epilogue)
    (2, 256, 64, 64, 256, 3, 1, 1),        # res4 3x3
    (2, 512, 128, 128, 1024, 1, 2, 0),     # strided 1x1 shortcut (scatter data gradient)
    (4000, 12544, 1, 1, 2048, 1, 1, 0),    # DAN fc1 at BASELINE size
    (300, 256, 14, 14, 256, 3, 1, 1),      # mask-head 3x3 on 300 pooled rois
], ids=["p2_3x3", "res4_1x1", "res4_3x3", "shortcut_s2", "dan_fc1", "mask_3x3"])
def test_full_size_adjointness(cuda, shape):
    """BASELINE-size layers, no oracle needed: forward, data gradient and weight gradient are three views of one
    trilinear form, so <conv(x,w), dy> = <x, dgrad(dy,w)> = <w, wgrad(dy,x)> (dots accumulated in fp64)."""
    N, C_, H, W, O, k, s, p = shape
    gen = torch.Generator(device=cuda).manual_seed(7)
    x = torch.randn(N, C_, H, W, device=cuda, generator=gen).contiguous(memory_format=CL)
    w = (torch.randn(O, C_, k, k, device=cuda, generator=gen) * (2.0 / (C_ * k * k)) ** 0.5).contiguous(memory_format=CL)
    y = K.conv2d_forward(x, w, s, p, 1)
    dy = torch.randn(y.shape, device=cuda, generator=gen).contiguous(memory_format=CL)
    dx = K.conv2d_backward_data(dy, w, tuple(x.shape), s, p, 1)
    dw = K.conv2d_backward_weight(dy, x, tuple(w.shape), s, p, 1)
    dot = lambda a, b: float((a.double() * b.double()).sum())
    f, d, g = dot(y, dy), dot(x, dx), dot(w, dw)
    scale = float(y.double().norm() * dy.double().norm())
    assert abs(f - d) <= REL * scale and abs(f - g) <= REL * scale, (f, d, g, scale)
    # and the forward is linear in x
    y2 = K.conv2d_forward(x * 0.5, w, s, p, 1)
    assert float((y2 - 0.5 * y).abs().max()) <= REL * float(y.abs().max())


def test_mask_head_3x3_on_many_pooled_rois(cuda, conv_math):
    """The mask heads' 3x3 convolutions on 14 x 14 pooled maps at a BASELINE-like roi count (the 256 x 256 tiling; with
    JTSM_X3_HALO_SMALL=1 the whole-image halo patches, csrc/conv_x3.h: x3_halo_ok).  Forward (fused epilogue) and data
    gradient (accumulate + gate) against the torch-CPU oracle."""
    N, C_, O = 176, 256, 256
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, C_, 14, 14, generator=g)
    w = torch.randn(O, C_, 3, 3, generator=g) * (2.0 / (C_ * 9)) ** 0.5
    bias = torch.randn(O, generator=g)
    xd, wd = x.to(cuda).contiguous(memory_format=CL), w.to(cuda).contiguous(memory_format=CL)
    y = K.conv2d_forward(xd, wd, 1, 1, 1, None, bias.to(cuda), None, True)
    close(y, nnref.conv_bn_act(x, w, 1, 1, 1, None, bias, None, True), "forward")
    dy = torch.randn(y.shape, generator=g)
    acc, gate = torch.randn(x.shape, generator=g), torch.randn(x.shape, generator=g)
    xr = x.clone().requires_grad_(True)
    nnref.conv_bn_act(xr, w, 1, 1, 1).backward(dy)
    dyd = dy.to(cuda).contiguous(memory_format=CL)
    dx = K.conv2d_backward_data(dyd, wd, tuple(x.shape), 1, 1, 1, accumulate=acc.to(cuda), relu_mask=gate.to(cuda))
    close(dx, (xr.grad + acc) * (gate > 0), "dgrad+acc+gate")


def test_splitk_finishing_inside_the_kernel_is_bit_identical_to_the_separate_pass(cuda, conv_math):
    """Split-K layers (few output tiles, long K) finished by the tile's last-arriving slice — slabs published with sc1
    stores, read back with sc1 loads, an agent-scope ticket, no device-scope fence — must give the very bits of the
    separate splitk_finish pass (same slice order, same epilogue).  Forward / data gradient / weight gradient (generic,
    LDS-halo and 256x256-tile forms), repeated: a lost or stale slab would show up as a differing element."""
    if conv_math == "f32":
        pytest.skip("the exact-fp32 kernels keep the separate finishing pass")
    from jtsm_amd import _lib as L
    gen = torch.Generator(device=cuda).manual_seed(12)
    cases = [(2, 256, 32, 32, 256, 3, 1, 1),     # 3x3 halo forms (few tiles at 32x32, split over channel blocks)
             (2, 1024, 16, 16, 256, 1, 1, 0),    # 1x1, K = 1024: 128x128 tiles, split
             (2, 512, 16, 16, 2048, 1, 1, 0),    # wide output
             (1, 256, 64, 64, 256, 3, 1, 1),     # wgrad halo with many pixel slices
             (2, 2048, 8, 8, 512, 1, 1, 0)]
    lib = L.lib()
    try:
        for (n, c, h, w, o, k, s, p) in cases:
            x = torch.randn(n, c, h, w, device=cuda, generator=gen).contiguous(memory_format=CL)
            wt = (torch.randn(o, c, k, k, device=cuda, generator=gen) * 0.05).contiguous(memory_format=CL)
            sc, bi = torch.rand(o, device=cuda, generator=gen) + 0.5, torch.randn(o, device=cuda, generator=gen)
            outs = {}
            for mode in (0, 1, 1, 1):
                lib.jtsm_conv_set_splitk_fused(mode)
                K.planes_clear()
                y = K.conv2d_forward(x, wt, s, p, 1, sc, bi, None, True, emit_planes=True)
                dy = torch.randn(y.shape, device=cuda, generator=torch.Generator(device=cuda).manual_seed(3)).contiguous(memory_format=CL)
                dx = K.conv2d_backward_data(dy, wt, tuple(x.shape), s, p, 1, kscale=sc, relu_mask=x)
                dw = K.conv2d_backward_weight(dy, x, tuple(wt.shape), s, p, 1, row_scale=sc)
                got = (y, dx, dw, K.planes_of(y).clone())
                if mode == 0:
                    outs = got
                else:
                    for a, b, what in zip(got, outs, ("y", "dx", "dw", "planes of y")):
                        assert torch.equal(a, b), ((n, c, h, w, o, k), what, float((a.float() - b.float()).abs().max()))
    finally:
        lib.jtsm_conv_set_splitk_fused(-1)


def test_paired_weight_planes_are_the_separate_planes_rearranged(cuda, conv_math):
    """Weights go to the contractions as PAIRED planes (csrc/conv_x3.h `x3_paired`: per row, blocks of 32 hi values
    followed by their 32 lo values).  Bit for bit the same numbers as the two separate planes, for the straight split,
    the transposing split, and the one-launch re-split of every cached weight after an update."""
    if conv_math != "bf16x3":
        pytest.skip("paired planes exist in the split-bf16 arithmetic only")
    assert K.W_PAIRED

    def unpair(buf, rows, k):
        b = buf[:2 * rows * k].view(rows, k // 32, 2, 32)
        return b[:, :, 0].reshape(rows, k), b[:, :, 1].reshape(rows, k)

    g = torch.Generator().manual_seed(5)
    for (o, i, kh) in [(48, 64, 3), (80, 256, 1), (256, 32, 1), (8, 96, 3)]:
        w = torch.nn.Parameter(torch.randn(o, i, kh, kh, generator=g).cuda().contiguous(memory_format=CL))
        hi, lo = K.split_bf16(w.detach())                                           # [out][taps][in], separate planes
        buf = K._weight_planes(w)
        assert getattr(buf, "_paired", False)
        ph, pl = unpair(buf, o, kh * kh * i)
        assert torch.equal(ph.reshape(-1), hi) and torch.equal(pl.reshape(-1), lo)
        scale = torch.rand(o, generator=g).cuda() + 0.5
        th, tl = K.split_bf16_transposed(w.detach(), scale)                         # [in][taps][out], separate planes
        tbuf = K._weight_planes(w, True, scale)
        if (kh * kh * o) % 32 == 0:
            assert getattr(tbuf, "_paired", False)
            ph, pl = unpair(tbuf, i, kh * kh * o)
            assert torch.equal(ph.reshape(-1), th) and torch.equal(pl.reshape(-1), tl)
        else:
            assert not getattr(tbuf, "_paired", False)
        # an update, then the table-driven re-split of all stale entries
        with torch.no_grad():
            w.mul_(1.7)
        K.refresh_weight_planes()
        hi2, lo2 = K.split_bf16(w.detach())
        ph, pl = unpair(K._weight_planes(w), o, kh * kh * i)
        assert torch.equal(ph.reshape(-1), hi2) and torch.equal(pl.reshape(-1), lo2)
        th2, tl2 = K.split_bf16_transposed(w.detach(), scale)
        tb2 = K._weight_planes(w, True, scale)
        if getattr(tb2, "_paired", False):
            ph, pl = unpair(tb2, i, kh * kh * o)
            assert torch.equal(ph.reshape(-1), th2) and torch.equal(pl.reshape(-1), tl2)
        assert not torch.equal(hi2, hi)


@pytest.mark.parametrize("tap", [False, True], ids=["chain", "tapped"])
def test_block_output_gate_rides_in_the_next_blocks_epilogue(cuda, monkeypatch, tap):
    """Inside a stage the ReLU gate of block k-1's output gradient is applied by block k's conv1 data-gradient epilogue
    (layers/fused_blocks.py: _PREGATED): same bits as the separate relu_backward pass, two passes fewer over a chain of
    three blocks.  `tapped`: the middle activation has a second consumer, so the gradient block 0 receives is a SUM —
    it must fail the identity check and be gated the ordinary way (gating twice is harmless, skipping it is not)."""
    from jtsm_amd.layers import fused_blocks as FB
    from jtsm_amd.modeling.backbone.resnet import BottleneckBlock
    torch.manual_seed(5)
    blocks = [BottleneckBlock(64, 64, bottleneck_channels=32, norm="FrozenBN").to(cuda) for _ in range(3)]
    for b in blocks:   # FrozenBN away from the identity, so that pre-activations take both signs
        for c in (b.conv1, b.conv2, b.conv3):
            c.norm.bias.copy_(torch.randn_like(c.norm.bias) * 0.3)
            c.norm.weight.copy_(torch.rand_like(c.norm.weight) + 0.5)
    x0 = torch.randn(2, 64, 24, 20, device=cuda).contiguous(memory_format=CL)
    wgt = torch.randn(2, 64, 24, 20, device=cuda).contiguous(memory_format=CL)
    wgt2 = torch.randn(2, 64, 24, 20, device=cuda).contiguous(memory_format=CL)
    real_relu_backward = FB.relu_backward

    def run(pregate):
        monkeypatch.setattr(FB, "PREGATE", pregate)
        K.planes_clear()
        for b in blocks:
            b.zero_grad(set_to_none=True)
        calls = []

        def counted(*a, **k):
            calls.append(1)
            return real_relu_backward(*a, **k)

        monkeypatch.setattr(FB, "relu_backward", counted)
        x = x0.clone().requires_grad_()
        y1 = blocks[0](x)
        y2 = blocks[1](y1)
        y3 = blocks[2](y2)
        loss = (y3 * wgt).sum()
        if tap:
            loss = loss + (y1 * wgt2).sum()
        loss.backward()
        monkeypatch.setattr(FB, "relu_backward", real_relu_backward)
        assert FB._PREGATED[0] is None or not pregate or tap      # every handed tensor was taken
        grads = [x.grad.clone()] + [p.grad.clone() for b in blocks for p in b.parameters() if p.grad is not None]
        return grads, len(calls)

    ref, n_ref = run(False)
    got, n_got = run(True)
    assert n_ref == 3 and n_got == (2 if tap else 1), (n_ref, n_got)
    assert len(ref) == len(got) == 1 + 9
    assert torch.equal(got[0], ref[0]), float((got[0] - ref[0]).abs().max())     # the chain's data gradient: same bits
    for a, b in zip(got[1:], ref[1:]):
        if K.MATH == "f32":   # the exact-fp32 weight gradient folds its K slices with atomics: not reproducible to the bit
            assert float((a - b).abs().max()) <= 1e-6 * float(b.abs().max())
        else:
            assert torch.equal(a, b), float((a - b).abs().max())


@pytest.mark.parametrize("math", ["bf16x3", "f16"])
def test_deferred_weight_gradients_are_grouped_and_match_the_single_launches(cuda, math):
    """layers/conv.py: with DEFER_WGRAD the fused bottleneck node queues its weight gradients and the same-shape layers
    of a stage go out as ONE grouped launch (jtsm_conv2d_backward_weight_group_*).  A stage of one projection block +
    four identity blocks: the data gradient is the same bits either way; every weight gradient agrees with the
    single-launch result to summation order (different K slicing: 1e-5 of the tensor's magnitude; fp16 planes round the
    same operands the same way, so the bar is the same), twice in a row gives the same bits, and the launch log shows
    the groups (conv1 x4, conv2 x4, conv3 x5 -> 3 grouped launches + 3 single ones instead of 16)."""
    from jtsm_amd.modeling.backbone.resnet import BottleneckBlock
    old = K.MATH
    K.set_math(math)
    try:
        torch.manual_seed(11)
        blocks = [BottleneckBlock(64, 128, bottleneck_channels=32, stride=2, norm="FrozenBN").to(cuda)] + \
                 [BottleneckBlock(128, 128, bottleneck_channels=32, norm="FrozenBN").to(cuda) for _ in range(4)]
        for b in blocks:
            for c in [b.conv1, b.conv2, b.conv3] + ([b.shortcut] if b.shortcut is not None else []):
                c.norm.bias.copy_(torch.randn_like(c.norm.bias) * 0.3)
                c.norm.weight.copy_(torch.rand_like(c.norm.weight) + 0.5)
        x0 = torch.randn(2, 64, 64, 96, device=cuda).contiguous(memory_format=CL)
        # (small output gradients: the fp16 planes carry them times 2^12, and |g| 2^12 must stay below fp16's 65504)
        wgt = (torch.randn(2, 128, 32, 48, device=cuda) * 1e-2).contiguous(memory_format=CL)

        def run(defer, log=False):
            K.defer_weight_gradients(defer)
            K.planes_clear()
            for b in blocks:
                b.zero_grad(set_to_none=True)
            x = x0.clone().requires_grad_()
            y = x
            for b in blocks:
                y = b(y)
            K.LAUNCH_LOG = [] if log else None
            (y * wgt).sum().backward()
            names, K.LAUNCH_LOG = [str(e[0]) for e in (K.LAUNCH_LOG or [])], None
            params = [p for b in blocks for p in b.parameters()]
            assert all(p.grad is not None for p in params)
            return x.grad.clone(), [p.grad.clone() for p in params], names

        dx0, g0, _ = run(False)
        dx1, g1, names = run(True, log=True)
        dx2, g2, _ = run(True)
        assert torch.equal(dx0, dx1) and torch.equal(dx1, dx2)
        assert sum("group" in n for n in names) == 3, names
        for a, b, c in zip(g0, g1, g2):
            assert torch.equal(b, c)                                            # reproducible
            assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()), float((a - b).abs().max())
        # the queued launches on the side stream (layers/conv.py: WGRAD_STREAM): the same kernels on the same operands
        # beside the backward instead of inside it — the same bits, and the compute stream has waited for them when
        # backward() returns (the gradients are read here without any synchronisation of ours)
        old_side = K.WGRAD_STREAM
        K.WGRAD_STREAM = True
        try:
            dx3, g3, _ = run(True)
            dx4, g4, _ = run(True)
        finally:
            K.WGRAD_STREAM = old_side
        assert torch.equal(dx3, dx1) and torch.equal(dx4, dx1)
        for b, c, d in zip(g1, g3, g4):
            assert torch.equal(b, c) and torch.equal(b, d)
        # accumulation into an existing .grad (no zero_grad between two backward passes): the sum of both
        K.defer_weight_gradients(True)
        x = x0.clone().requires_grad_()
        y = x
        for b in blocks:
            y = b(y)
        (y * wgt).sum().backward()
        for p, b in zip([p for blk in blocks for p in blk.parameters()], g1):
            assert float((p.grad - 2 * b).abs().max()) <= 1e-5 * float(b.abs().max())
    finally:
        K.defer_weight_gradients(True)
        K.set_math(old)
