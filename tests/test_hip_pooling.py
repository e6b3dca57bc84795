"""GPU suite (pytest -m gpu): the HIP pooling kernels, called through the C ABI via the
jtsm_amd.layers mirrors, against the oracle and the committed reference vectors.

Bar: integer results (sample grid, tap indices, argmax, mois) bit-exact; forward VALUES are
also required bit-exact (same summation order, contraction off); backward uses float atomics
(order-dependent rounding) -> rtol 1e-5 / atol 1e-6 in fp32, 1e-12 in fp64.
"""
import numpy as np
import pytest
import torch

from conftest import load_cases
from oracle import pooling as P

pytestmark = pytest.mark.gpu

from jtsm_amd import _lib as L  # noqa: E402
from jtsm_amd.layers import MOIPool, ROIAlign, ROIAlignRotated  # noqa: E402

KAT = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "kat_reference_tests.npz"))


def dev(a, cuda, channels_last=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    if channels_last:
        t = t.contiguous(memory_format=torch.channels_last)
    return t


def run_align(x, rois, scale, PH, PW, sr, aligned, cuda, nhwc, g=None, rotated=False):
    xt = dev(x, cuda, nhwc).requires_grad_(g is not None)
    op = ROIAlignRotated((PH, PW), scale, sr) if rotated else ROIAlign((PH, PW), scale, sr, aligned)
    y = op(xt, dev(rois, cuda))
    if nhwc and y.numel():
        assert y.is_contiguous(memory_format=torch.channels_last)
    gx = None
    if g is not None:
        y.backward(dev(g, cuda, nhwc))
        gx = xt.grad.cpu().numpy()
    return y.detach().cpu().numpy(), gx


@pytest.mark.parametrize("nhwc", [False, True])
def test_kat_tables(cuda, nhwc):
    img = KAT["image5x5"][None, None]
    box = np.array([[0, *KAT["box"]]], np.float32)
    y, _ = run_align(img, box, 1.0, 4, 4, 0, False, cuda, nhwc)
    assert np.allclose(y[0, 0], KAT["legacy_4x4"])
    y, _ = run_align(img, box, 1.0, 4, 4, 0, True, cuda, nhwc)
    assert np.allclose(y[0, 0], KAT["aligned_4x4"])
    b = KAT["box"]
    for i in range(4):
        r = np.array([[0, (b[0] + b[2]) / 2, (b[1] + b[3]) / 2, b[2] - b[0], b[3] - b[1], 90 * i]], np.float32)
        y, _ = run_align(img, r, 1.0, 4, 4, 0, True, cuda, nhwc, rotated=True)
        exp = KAT["aligned_4x4"]
        for _ in range((-i) % 4):
            exp = exp.T[::-1]
        assert np.allclose(y[0, 0], exp), i


@pytest.mark.parametrize("nhwc", [False, True])
def test_empty_box_and_empty_batch(cuda, nhwc):
    img = np.random.default_rng(0).random((1, 1, 5, 5)).astype(np.float32)
    y, gx = run_align(img, np.array([[0, *KAT["empty_box"]]], np.float32), 1.0, 7, 7, 0, True, cuda, nhwc,
                      g=np.ones((1, 1, 7, 7), np.float32))
    assert (y == 0).all() and (gx == 0).all()
    y, _ = run_align(img, np.array([[0, *KAT["rotated_empty_box"]]], np.float32), 1.0, 7, 7, 0, True, cuda,
                     nhwc, rotated=True)
    assert (y == 0).all()
    op = ROIAlign((7, 7), 1.0, 0, aligned=True)
    out = op(torch.zeros(0, 3, 10, 10, device=cuda), torch.zeros(0, 5, device=cuda))
    assert out.shape == (0, 3, 7, 7)
    out = op(torch.zeros(2, 3, 10, 10, device=cuda), torch.zeros(0, 5, device=cuda))
    assert out.shape == (0, 3, 7, 7)


@pytest.mark.parametrize("nhwc", [False, True])
@pytest.mark.parametrize("case", sorted(load_cases("roi_align_ref.npz")))
def test_roi_align_vs_reference_vectors(cuda, case, nhwc):
    c = load_cases("roi_align_ref.npz")[case]
    scale, PH, PW, sr, al = c["meta"]
    y, gx = run_align(c["x"], c["rois"], float(scale), int(PH), int(PW), int(sr), bool(al), cuda, nhwc,
                      g=c["g"])
    assert np.array_equal(y, c["y"]), np.abs(y - c["y"]).max()
    tol = dict(rtol=1e-5, atol=1e-5) if c["x"].dtype == np.float32 else dict(rtol=1e-12, atol=1e-12)
    assert np.allclose(gx, c["gx"], **tol), np.abs(gx - c["gx"]).max()


@pytest.mark.parametrize("nhwc", [False, True])
@pytest.mark.parametrize("case", sorted(load_cases("roi_align_rotated_ref.npz")))
def test_roi_align_rotated_vs_reference_vectors(cuda, case, nhwc):
    c = load_cases("roi_align_rotated_ref.npz")[case]
    scale, PH, PW, sr = c["meta"]
    y, gx = run_align(c["x"], c["rois"], float(scale), int(PH), int(PW), int(sr), True, cuda, nhwc,
                      g=c["g"], rotated=True)
    assert np.array_equal(y, c["y"]), np.abs(y - c["y"]).max()
    tol = dict(rtol=1e-5, atol=1e-5) if c["x"].dtype == np.float32 else dict(rtol=1e-12, atol=1e-12)
    assert np.allclose(gx, c["gx"], **tol)


def _fpn_like_rois(rng, M, B, size):
    x0, y0 = rng.uniform(0, size * 0.75, M), rng.uniform(0, size * 0.75, M)
    w = np.exp(rng.uniform(np.log(16), np.log(size / 2), M))
    h = np.exp(rng.uniform(np.log(16), np.log(size / 2), M))
    return np.stack([rng.integers(0, B, M), x0, y0, np.minimum(x0 + w, size), np.minimum(y0 + h, size)],
                    1).astype(np.float32)


@pytest.mark.parametrize("rotated", [False, True])
def test_sample_tables_bit_exact(cuda, rotated):
    """The 'bit-exact ROI bin indices' contract: grid sizes, the 4 tap indices and the 4 weights of
    every sample equal the oracle's, for boxes spanning all FPN scales incl. borders."""
    rng = np.random.default_rng(1234)
    H, W, scale, PH, PW = 64, 80, 1.0 / 16, 7, 7
    M = 256
    r = _fpn_like_rois(rng, M, 1, 1024)
    r[:8, 1:] = [[-40, -40, 30, 30], [1000, 990, 1300, 1400], [0, 0, 1280, 1024], [5, 5, 5, 5],
                 [100.5, 200.25, 100.5, 900], [16, 16, 32, 32], [15.999, 16.001, 240, 33], [3, 4, 5, 4]]
    if rotated:
        r = np.concatenate([r[:, :1], (r[:, 1:3] + r[:, 3:5]) / 2, r[:, 3:5] - r[:, 1:3],
                            rng.uniform(-180, 180, (M, 1)).astype(np.float32)], 1).astype(np.float32)
        r[:4, 5] = [0, 90, 180, 270]
    cap = 7 * 7 * 36
    grid = torch.empty(M, 2, dtype=torch.int32, device=cuda)
    pos = torch.empty(M, cap, 4, dtype=torch.int32, device=cuda)
    w = torch.empty(M, cap, 4, dtype=torch.float32, device=cuda)
    rd = dev(r, cuda)  # keep alive: a temporary would hand its block back to the allocator
    L.check(L.lib().jtsm_roi_sample_table_f32(L.ptr(rd), int(rotated), M, H, W, L.f32(scale), PH,
                                              PW, 0, 1, L.ptr(grid), L.ptr(pos), L.ptr(w), cap, L.stream()))
    grid, pos, w = grid.cpu().numpy(), pos.cpu().numpy(), w.cpu().numpy()
    checked = 0
    for m in range(M):
        g0, p0, w0 = P.roi_sample_table(r[m], rotated, H, W, scale, PH, PW, 0, True)
        assert np.array_equal(grid[m], g0), (m, grid[m], g0)
        n = min(len(p0), cap)
        assert np.array_equal(pos[m, :n], p0[:n]), m
        assert np.array_equal(w[m, :n].view(np.uint32), w0[:n].view(np.uint32)), m
        checked += n
    assert checked > 30000


@pytest.mark.parametrize("nhwc", [False, True])
def test_config2_shape_forward_bit_exact_and_backward_close(cuda, nhwc):
    """BASELINE configs[1] geometry on one FPN level (p3: 128x128, C=256) with 64 rois:
    values bit-exact vs the oracle, 7x7 and 14x14."""
    rng = np.random.default_rng(99)
    B, Cc, H, W = 2, 256, 128, 128
    x = rng.standard_normal((B, Cc, H, W)).astype(np.float32)
    r = _fpn_like_rois(rng, 64, B, 1024)
    for res in (7, 14):
        g = rng.standard_normal((64, Cc, res, res)).astype(np.float32)
        y, gx = run_align(x, r, 0.125, res, res, 0, True, cuda, nhwc, g=g)
        y0 = P.roi_align_forward(x, r, 0.125, res, res, 0, True)
        assert np.array_equal(y, y0)
        gx0 = P.roi_align_backward(g, r, 0.125, res, res, B, Cc, H, W, 0, True)
        assert np.allclose(gx, gx0, rtol=1e-4, atol=1e-4), np.abs(gx - gx0).max()


def test_full_size_properties(cuda):
    """configs[1] at full size (8 x 256 x 256 x 256 NHWC, 512 rois): properties that need no oracle.
    (1) linearity in the input, (2) a constant map pools to that constant where all samples are
    inside, (3) NHWC and NCHW kernels agree bit for bit, (4) sum(backward(1)) == #valid samples /
    count per bin == number of non-empty bins."""
    rng = np.random.default_rng(5)
    B, Cc, H, W, M = 8, 256, 256, 256, 512
    gen = torch.Generator(device=cuda).manual_seed(0)
    x = torch.randn(B, Cc, H, W, device=cuda, generator=gen).contiguous(memory_format=torch.channels_last)
    r_np = _fpn_like_rois(rng, M, B, 1024)
    r = torch.from_numpy(r_np).to(cuda)
    op = ROIAlign((7, 7), 0.25, 0, aligned=True)
    y = op(x, r)
    y2 = op(2.5 * x, r)
    assert torch.allclose(y2, 2.5 * y, rtol=1e-5, atol=1e-5)
    ones = torch.ones_like(x)
    yc = op(ones, r)
    inside = (r[:, 1] > 4) & (r[:, 2] > 4) & (r[:, 3] < 1016) & (r[:, 4] < 1016)
    assert torch.allclose(yc[inside], torch.ones_like(yc[inside]), atol=1e-5)
    sub = slice(0, 64)
    y_nchw = op(x.contiguous(), r[sub])
    assert torch.equal(y_nchw, y[sub].contiguous())
    xg = torch.ones(1, 4, H, W, device=cuda).contiguous(memory_format=torch.channels_last).requires_grad_()
    r1 = r.clone()
    r1[:, 0] = 0
    op(xg, r1).sum().backward()
    per_roi_bins = 49.0
    assert abs(xg.grad.sum().item() / 4 - M * per_roi_bins) < 1e-2 * M


# ----------------------------------------------------------------------------- MOIPool
def run_moi(c, cuda, nhwc, scale, PH, PW, g=None):
    x = dev(c["x"], cuda, nhwc).requires_grad_(g is not None)
    op = MOIPool((PH, PW), scale)
    y, a = op(x, dev(c["rois"], cuda), dev(c["oh"], cuda), dev(c["sp"], cuda))
    gx = None
    if g is not None:
        y.backward(dev(g, cuda, nhwc))
        gx = x.grad.cpu().numpy()
    return y.detach().cpu().numpy(), a.cpu().numpy(), gx


@pytest.mark.parametrize("nhwc", [False, True])
@pytest.mark.parametrize("case", sorted(load_cases("moi_pool_oracle.npz")))
def test_moi_pool_vs_oracle_vectors(cuda, case, nhwc):
    c = load_cases("moi_pool_oracle.npz")[case]
    scale, PH, PW = c["meta"]
    y, a, gx = run_moi(c, cuda, nhwc, float(scale), int(PH), int(PW), g=c["g"])
    assert np.array_equal(a, c["argmax"])
    assert np.array_equal(y, c["y"])
    assert np.allclose(gx, c["gx"], rtol=1e-5, atol=1e-5)


def test_moi_mask_bit_exact(cuda):
    c = load_cases("moi_pool_oracle.npz")["a_stride4"]
    B, Cc, H, W = c["x"].shape
    M, Lw = c["oh"].shape
    Hs, Ws = c["sp"].shape[1:]
    mois = torch.empty(M, H, W, dtype=torch.int32, device=cuda)
    ws = torch.empty(L.lib().jtsm_moi_pool_workspace_bytes(B, H, W, M, Lw), dtype=torch.uint8, device=cuda)
    rd, ohd, spd = dev(c["rois"], cuda), dev(c["oh"], cuda), dev(c["sp"], cuda)  # keep alive
    L.check(L.lib().jtsm_moi_mask_f32(L.ptr(rd), L.ptr(ohd), L.ptr(spd), L.ptr(mois), L.ptr(ws), B, H, W, M,
                                      Lw, Hs, Ws, L.f32(0.25), L.stream()))
    assert np.array_equal(mois.cpu().numpy(), P.moi_mask(c["rois"], c["oh"], c["sp"], H, W, 0.25))


@pytest.mark.parametrize("nhwc", [False, True])
def test_moi_pool_jtsm_shape(cuda, nhwc):
    """configs[2] geometry on one level (p3, C=256, 1024 superpixels of 32x32 px, 200 rois):
    argmax and values bit-exact vs the oracle; >32 label words exercises the multi-word path."""
    rng = np.random.default_rng(77)
    B, Cc, H, W, stride = 2, 256, 64, 64, 8
    Hs = Ws = H * stride
    ids = (np.arange(Hs)[:, None] // 16) * (Ws // 16) + (np.arange(Ws)[None, :] // 16)
    sp = np.stack([ids, np.roll(ids, (5, 9), (0, 1))]).astype(np.int32)
    Lw = int(ids.max()) + 1
    assert Lw == 1024
    M = 200
    r = _fpn_like_rois(rng, M, B, Hs)
    cy, cx = (np.arange(Hs // 16) * 16 + 8), (np.arange(Ws // 16) * 16 + 8)
    oh = np.zeros((M, Lw), np.int32)
    for m in range(M):
        iny = (cy >= r[m, 2]) & (cy <= r[m, 4])
        inx = (cx >= r[m, 1]) & (cx <= r[m, 3])
        oh[m] = (iny[:, None] & inx[None, :]).ravel()
    x = rng.standard_normal((B, Cc, H, W)).astype(np.float32)
    c = dict(x=x, rois=r, oh=oh, sp=sp)
    g = rng.standard_normal((M, Cc, 7, 7)).astype(np.float32)
    y, a, gx = run_moi(c, cuda, nhwc, 1.0 / stride, 7, 7, g=g)
    y0, a0 = P.moi_pool_forward(x, r, 1.0 / stride, 7, 7, oh, sp)
    assert np.array_equal(a, a0) and np.array_equal(y, y0)
    gx0 = P.moi_pool_backward(g, r, a0, 1.0 / stride, 7, 7, B, Cc, H, W)
    assert np.allclose(gx, gx0, rtol=1e-4, atol=1e-4)
    assert (a0 >= 0).mean() > 0.3  # the case is not degenerate


def test_fpn_level_assignment_and_multilevel_pooling_match_oracle(cuda):
    """the integer FPN level of every box (floor(4 + log2(sqrt(area)/224 + 1e-8)),
    clamped to [2,5]) computed on the device equals the CPU restatement — including boxes sitting exactly
    on the level boundaries (sqrt(area) = 112, 224, 448) — and the sync-free multi-level launches
    (one launch per level over ALL boxes) reproduce the oracle's per-level gather/pool/scatter."""
    from jtsm_amd.modeling.poolers import ROIPooler, assign_boxes_to_levels
    from jtsm_amd.structures import Boxes
    from oracle import model as OM

    rng = np.random.default_rng(31)
    M = 600
    r = _fpn_like_rois(rng, M, 2, 1024)
    edge = np.array([[0, 10, 10, 10 + s, 10 + s] for s in (112, 224, 448, 111.99999, 224.00002, 447.9999, 896)],
                    np.float32)
    r[:len(edge)] = edge
    boxes = [Boxes(torch.from_numpy(r[r[:, 0] == b][:, 1:]).to(cuda)) for b in (0, 1)]
    lv = assign_boxes_to_levels(boxes, 2, 5, 224, 4).cpu()
    cat = torch.cat([torch.from_numpy(r[r[:, 0] == b][:, 1:]) for b in (0, 1)])
    lv0 = OM.assign_levels(cat)
    assert torch.equal(lv, lv0)
    assert set(lv.tolist()) == {0, 1, 2, 3}

    feats = [rng.standard_normal((2, 8, 256 >> i, 256 >> i)).astype(np.float32) for i in range(4)]
    pooler = ROIPooler(7, [1 / 4, 1 / 8, 1 / 16, 1 / 32], 0, "ROIAlignV2")
    xs = [dev(f, cuda, True).requires_grad_() for f in feats]
    y = pooler(xs, boxes)
    g = rng.standard_normal(tuple(y.shape)).astype(np.float32)
    y.backward(dev(g, cuda, True))
    fr = [torch.from_numpy(f).requires_grad_() for f in feats]
    rois = torch.cat([torch.cat([torch.full((len(b), 1), float(i)), b.tensor.cpu()], 1) for i, b in enumerate(boxes)])
    y0 = OM.roi_align_levels(fr, rois, 1024, 7)
    y0.backward(torch.from_numpy(g))
    assert np.array_equal(y.detach().cpu().numpy(), y0.detach().numpy())
    for a, b in zip(xs, fr):
        assert np.allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=1e-4, atol=1e-5)


def test_multilevel_roi_align_rotated_pooler_matches_oracle_and_v2_at_zero_degrees(cuda):
    """ROIPooler's level loop with pooler_type "ROIAlignRotated" (detectron2/modeling/poolers.py:160-165,230-249) on a
    4-level pyramid: (1) rotated boxes at random angles — the level of each box from width x height
    (RotatedBoxes.area), forward bit-exact against the C oracle's rotated kernel run level by level (gather / pool /
    scatter as the reference does), backward 1e-5; (2) the reference's own pooler test
    (tests/modeling/test_roi_pooler.py:15-60): ROIAlignV2 on (x0,y0,x1,y1) boxes equals ROIAlignRotated on the same
    boxes as (cx,cy,w,h,0 deg) within its atol 1e-4 — here on four levels, forward and backward."""
    from jtsm_amd.modeling.poolers import ROIPooler, assign_boxes_to_levels, convert_boxes_to_pooler_format
    from jtsm_amd.structures import Boxes, RotatedBoxes
    from oracle import model as OM

    rng = np.random.default_rng(77)
    M, B, Cc, res = 400, 2, 8, 7
    r = _fpn_like_rois(rng, M, B, 1024)
    scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
    feats = [rng.standard_normal((B, Cc, 256 >> i, 256 >> i)).astype(np.float32) for i in range(4)]

    def rot(r, angles):
        return np.concatenate([(r[:, 1:3] + r[:, 3:5]) / 2, r[:, 3:5] - r[:, 1:3], angles[:, None]], 1).astype(np.float32)

    ang = rng.uniform(-180, 180, M).astype(np.float32)
    ang[:4] = [0, 90, 180, 270]
    rb = rot(r, ang)
    rboxes = [RotatedBoxes(torch.from_numpy(rb[r[:, 0] == b]).to(cuda)) for b in range(B)]
    order = np.concatenate([np.nonzero(r[:, 0] == b)[0] for b in range(B)])
    r, rb = r[order], rb[order]                                    # rows in the pooler's (image-major) order
    lv = assign_boxes_to_levels(rboxes, 2, 5, 224, 4).cpu()
    assert torch.equal(lv, OM.assign_levels(torch.from_numpy(r[:, 1:])))     # same w x h -> same level
    assert set(lv.tolist()) == {0, 1, 2, 3}
    fmt = convert_boxes_to_pooler_format(rboxes)
    assert tuple(fmt.shape) == (M, 6) and np.array_equal(fmt.cpu().numpy()[:, 1:], rb)

    pooler = ROIPooler(res, scales, 0, "ROIAlignRotated")
    xs = [dev(f, cuda, True).requires_grad_() for f in feats]
    y = pooler(xs, rboxes)
    g = rng.standard_normal(tuple(y.shape)).astype(np.float32)
    y.backward(dev(g, cuda, True))
    rois6 = np.concatenate([r[:, :1], rb], 1).astype(np.float32)
    y0 = np.zeros((M, Cc, res, res), np.float32)
    for l, f in enumerate(feats):
        sel = np.nonzero(lv.numpy() == l)[0]
        y0[sel] = P.roi_align_rotated_forward(f, rois6[sel], scales[l], res, res, 0)
        gx0 = P.roi_align_rotated_backward(np.ascontiguousarray(g[sel]), rois6[sel], scales[l], res, res, B, Cc,
                                           f.shape[2], f.shape[3], 0)
        gx = xs[l].grad.cpu().numpy()
        assert np.allclose(gx, gx0, rtol=1e-5, atol=1e-5 * np.abs(gx0).max()), (l, np.abs(gx - gx0).max())
    assert np.array_equal(y.detach().cpu().numpy(), y0)

    # (2) V2 == rotated at 0 degrees
    boxes = [Boxes(torch.from_numpy(r[r[:, 0] == b][:, 1:]).to(cuda)) for b in range(B)]
    zero = [RotatedBoxes(torch.from_numpy(rot(r[r[:, 0] == b], np.zeros(int((r[:, 0] == b).sum()), np.float32))).to(cuda))
            for b in range(B)]
    out = []
    for kind, bl in (("ROIAlignV2", boxes), ("ROIAlignRotated", zero)):
        xs = [dev(f, cuda, True).requires_grad_() for f in feats]
        yk = ROIPooler(res, scales, 0, kind)(xs, bl)
        yk.backward(dev(g, cuda, True))
        out.append((yk.detach().cpu().numpy(), [x.grad.cpu().numpy() for x in xs]))
    assert np.allclose(out[0][0], out[1][0], atol=1e-4)
    for a, b in zip(out[0][1], out[1][1]):
        assert np.allclose(a, b, atol=1e-4 * max(1.0, np.abs(a).max()))


@pytest.mark.parametrize("sp,size", [(8, 256), (16, 256), (4, 128), (32, 256), (12, 240), (8, 360), (32, 352)])
def test_multilevel_moi_pool_forward_label_widths_and_bit_table_forms(cuda, sp, size):
    """The multi-level MOIPool forward over the label-set widths the per-(roi, bin row) kernel is instantiated for
    (ceil(L / 32) = 4, 8, 16, 32, 64 words; other widths take the per-(roi, bin) kernel) and over the three ways the cell
    bit tables are built (thread per cell at power-of-two superpixel ratios <= 8, register rows per wavefront on coarse
    levels, the LDS form when the map is not a power-of-two multiple of the level — size 240 with 12-px blocks): values
    and arg-max bit-exact against the oracle, level by level."""
    from jtsm_amd.modeling.poolers import ROIPooler
    from jtsm_amd.structures import Boxes
    from oracle import model as OM

    rng = np.random.default_rng(sp * 1000 + size)
    B, Cc, M = 2, 64, 160
    r = _fpn_like_rois(rng, M, B, size)
    r[:, 1:] *= 1.0                                   # boxes in the size x size image
    boxes = [torch.from_numpy(r[r[:, 0] == b][:, 1:]) for b in range(B)]
    grid = size // sp
    ids = (torch.arange(size)[:, None] // sp) * grid + (torch.arange(size)[None, :] // sp)
    superpixels = ids.to(torch.int32)[None].repeat(B, 1, 1)
    cy = torch.arange(grid) * sp + sp / 2.0
    oh = []
    for bx in boxes:
        iny = (cy[None, :] >= bx[:, 1:2]) & (cy[None, :] <= bx[:, 3:4])
        inx = (cy[None, :] >= bx[:, 0:1]) & (cy[None, :] <= bx[:, 2:3])
        oh.append((iny[:, :, None] & inx[:, None, :]).reshape(len(bx), -1).to(torch.int32))
    feats = [rng.standard_normal((B, Cc, (size // 4) >> i, (size // 4) >> i)).astype(np.float32) for i in range(4)]
    scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
    pooler = ROIPooler(7, scales, 0, "MOIPool")
    y, arg = pooler([dev(f, cuda, True) for f in feats], [Boxes(b.to(cuda)) for b in boxes],
                    oh_labels_list=[o.to(cuda) for o in oh], superpixels=superpixels.to(cuda))
    y0, a0 = OM.moi_pool_levels([torch.from_numpy(f) for f in feats], boxes, oh, superpixels)
    assert np.array_equal(arg.cpu().numpy(), a0.numpy())
    assert np.array_equal(y.cpu().numpy(), y0.numpy())
    assert float((a0 >= 0).float().mean()) > 0.05


def test_multilevel_moi_pool_backward_gather_matches_oracle_and_scatter(cuda):
    """The gather form of the multi-level MOIPool backward (one workgroup per 8x8-cell tile, LDS accumulation in
    roi / bin order; 256-channel maps) against (1) the oracle's per-level scatter, (2) the library's own atomic
    scatter form (scales == NULL) on the same inputs, and (3) itself run twice — it must be bitwise reproducible."""
    import ctypes as C

    from jtsm_amd import _lib as L
    from jtsm_amd.modeling.poolers import ROIPooler
    from jtsm_amd.structures import Boxes
    from oracle import model as OM

    rng = np.random.default_rng(41)
    B, Cc, size, sp = 2, 256, 256, 8
    M = 300
    r = _fpn_like_rois(rng, M, B, size)
    boxes = [torch.from_numpy(r[r[:, 0] == b][:, 1:]) for b in range(B)]
    grid = size // sp
    ids = (torch.arange(size)[:, None] // sp) * grid + (torch.arange(size)[None, :] // sp)
    superpixels = ids.to(torch.int32)[None].repeat(B, 1, 1)
    cy = torch.arange(grid) * sp + sp / 2.0
    oh = []
    for bx in boxes:
        iny = (cy[None, :] >= bx[:, 1:2]) & (cy[None, :] <= bx[:, 3:4])
        inx = (cy[None, :] >= bx[:, 0:1]) & (cy[None, :] <= bx[:, 2:3])
        oh.append((iny[:, :, None] & inx[:, None, :]).reshape(len(bx), -1).to(torch.int32))
    feats = [rng.standard_normal((B, Cc, (size // 4) >> i, (size // 4) >> i)).astype(np.float32) for i in range(4)]
    scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]

    def run():
        pooler = ROIPooler(7, scales, 0, "MOIPool")
        xs = [dev(f, cuda, True).requires_grad_() for f in feats]
        y, arg = pooler(xs, [Boxes(b.to(cuda)) for b in boxes], oh_labels_list=[o.to(cuda) for o in oh],
                        superpixels=superpixels.to(cuda))
        return xs, y, arg

    xs, y, arg = run()
    g = rng.standard_normal(tuple(y.shape)).astype(np.float32)
    gd = dev(g, cuda, True)
    y.backward(gd)
    # (1) oracle
    fr = [torch.from_numpy(f).requires_grad_() for f in feats]
    y0, a0 = OM.moi_pool_levels(fr, boxes, oh, superpixels)
    y0.backward(torch.from_numpy(g))
    assert np.array_equal(y.detach().cpu().numpy(), y0.detach().numpy())
    assert np.array_equal(arg.cpu().numpy(), a0.numpy())
    for a, b in zip(xs, fr):
        assert np.allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=1e-5, atol=1e-5)
    assert sum(float(a.grad.abs().sum()) for a in xs) > 0
    # (2) the atomic scatter form of the same library entry
    from jtsm_amd.modeling.poolers import assign_boxes_to_levels, convert_boxes_to_pooler_format
    bl = [Boxes(b.to(cuda)) for b in boxes]
    rois = convert_boxes_to_pooler_format(bl)
    lv = assign_boxes_to_levels(bl, 2, 5, 224, 4).to(torch.int32)
    nl = 4
    grads = [torch.full_like(a.grad, float("nan")) for a in xs]
    Hs = (C.c_int * nl)(*[a.shape[2] for a in xs])
    Ws = (C.c_int * nl)(*[a.shape[3] for a in xs])
    ptrs = (C.c_void_p * nl)(*[t.data_ptr() for t in grads])
    gcl, acl = gd.contiguous(memory_format=torch.channels_last), arg.contiguous(memory_format=torch.channels_last)
    L.check(L.lib().jtsm_moi_pool_backward_levels_f32(L.ptr(gcl), L.ptr(rois), L.ptr(lv), L.ptr(acl), ptrs, Hs, Ws, None,
                                                      nl, B, Cc, rois.shape[0], 7, 7, 0, None, C.c_size_t(0), L.stream()),
            "scatter form")
    for a, s in zip(xs, grads):
        assert torch.allclose(a.grad, s, rtol=1e-5, atol=1e-5)
    # (3) bitwise reproducible
    xs2, y2, _ = run()
    y2.backward(gd)
    for a, b in zip(xs, xs2):
        assert torch.equal(a.grad, b.grad)


def test_multilevel_moi_pool_backward_falls_back_when_rois_pile_up(cuda):
    """More than 4000 (roi, bin) pairs on one 8x8-cell tile (1300 small rois on one spot): the gather only clears the maps and the float-atomic scatter does
    the work (csrc/moi_pool.hip: census).  Same gradients as the oracle either way."""
    from jtsm_amd.modeling.poolers import ROIPooler
    from jtsm_amd.structures import Boxes
    from oracle import model as OM

    rng = np.random.default_rng(43)
    B, Cc, size, sp = 1, 256, 128, 8
    M = 1300
    x0 = rng.uniform(40, 44, M).astype(np.float32)
    y0 = rng.uniform(40, 44, M).astype(np.float32)
    wh = rng.uniform(17, 30, (M, 2)).astype(np.float32)
    boxes = [torch.from_numpy(np.stack([x0, y0, x0 + wh[:, 0], y0 + wh[:, 1]], 1))]
    grid = size // sp
    ids = (torch.arange(size)[:, None] // sp) * grid + (torch.arange(size)[None, :] // sp)
    superpixels = ids.to(torch.int32)[None]
    oh = [torch.ones(M, grid * grid, dtype=torch.int32)]
    feats = [rng.standard_normal((B, Cc, (size // 4) >> i, (size // 4) >> i)).astype(np.float32) for i in range(4)]
    pooler = ROIPooler(7, [1 / 4, 1 / 8, 1 / 16, 1 / 32], 0, "MOIPool")
    xs = [dev(f, cuda, True).requires_grad_() for f in feats]
    y, arg = pooler(xs, [Boxes(boxes[0].to(cuda))], oh_labels_list=[oh[0].to(cuda)], superpixels=superpixels.to(cuda))
    g = rng.standard_normal(tuple(y.shape)).astype(np.float32)
    y.backward(dev(g, cuda, True))
    fr = [torch.from_numpy(f).requires_grad_() for f in feats]
    y0_, a0 = OM.moi_pool_levels(fr, boxes, oh, superpixels)
    y0_.backward(torch.from_numpy(g))
    assert np.array_equal(arg.cpu().numpy(), a0.numpy())
    for a, b in zip(xs, fr):
        ref = b.grad.numpy()
        assert np.allclose(a.grad.cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(ref).max())))
    assert float(xs[0].grad.abs().sum()) > 0


def test_roi_align_backward_gather_reproducible_and_fallback(cuda):
    """NHWC float32 backward with a 64-multiple channel count runs as a tiled gather (csrc/roi_align.hip:
    align_bwd_tiled): bitwise reproducible, equal to the oracle within rounding; with a pile of rois on one tile the
    census sends the call to the float-atomic scatter — same gradients either way."""
    rng = np.random.default_rng(7)
    B, Cc, H, W = 2, 64, 64, 64
    x = rng.standard_normal((B, Cc, H, W)).astype(np.float32)
    # (1) spread-out rois (>= 8192 bins: the gather's threshold), legacy and aligned sampling, fixed and adaptive grids
    r = _fpn_like_rois(rng, 192, B, 512)
    for aligned, sr in ((True, 0), (False, 0), (True, 2)):
        g = rng.standard_normal((192, Cc, 7, 7)).astype(np.float32)
        _, gx = run_align(x, r, 0.125, 7, 7, sr, aligned, cuda, True, g=g)
        _, gx2 = run_align(x, r, 0.125, 7, 7, sr, aligned, cuda, True, g=g)
        assert np.array_equal(gx, gx2)
        gx0 = P.roi_align_backward(g, r, 0.125, 7, 7, B, Cc, H, W, sr, aligned)
        assert np.allclose(gx, gx0, rtol=1e-4, atol=1e-4), np.abs(gx - gx0).max()
    # (2) 600 small rois on one spot: > 2000 estimated bins on a tile -> scatter form
    M = 600
    x0 = rng.uniform(200, 204, M).astype(np.float32)
    y0 = rng.uniform(100, 104, M).astype(np.float32)
    wh = rng.uniform(20, 40, (M, 2)).astype(np.float32)
    r2 = np.stack([np.zeros(M, np.float32), x0, y0, x0 + wh[:, 0], y0 + wh[:, 1]], 1)
    g = rng.standard_normal((M, Cc, 7, 7)).astype(np.float32)
    _, gx = run_align(x, r2, 0.125, 7, 7, 0, True, cuda, True, g=g)
    gx0 = P.roi_align_backward(g, r2, 0.125, 7, 7, B, Cc, H, W, 0, True)
    assert np.allclose(gx, gx0, rtol=1e-4, atol=1e-3 * max(1.0, float(np.abs(gx0).max()) / 10)), np.abs(gx - gx0).max()
    assert float(np.abs(gx[1]).max()) == 0.0            # nothing lands in the other image


_SCATTER_WORKER = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from jtsm_amd.modeling.poolers import ROIPooler
from jtsm_amd.structures import Boxes
d = np.load(sys.argv[2])
cuda = torch.device("cuda", 0)
CL = torch.channels_last
xs = [torch.from_numpy(d["f%d" % i]).to(cuda).contiguous(memory_format=CL).requires_grad_() for i in range(4)]
boxes = [Boxes(torch.from_numpy(d["b%d" % i]).to(cuda)) for i in range(2)]
y = ROIPooler(7, [1 / 4, 1 / 8, 1 / 16, 1 / 32], 0, "ROIAlignV2")(xs, boxes)
y.backward(torch.from_numpy(d["g"]).to(cuda).contiguous(memory_format=CL))
np.savez(sys.argv[3], **{"g%d" % i: x.grad.cpu().numpy() for i, x in enumerate(xs)})
"""


def test_multilevel_roi_align_backward_scatter_fallback_in_one_launch(cuda, tmp_path):
    """The census fallback of the multi-level ROIAlign backward — the float-atomic scatter over the whole level table in
    ONE launch (csrc/roi_align.hip: align_bwd_nhwc_levels) — forced by a census limit of 1 in a child process (the
    library reads JTSM_ALIGN_CENSUS_LIMIT once): same gradients as the oracle's per-level scatter."""
    import os
    import subprocess
    import sys

    from conftest import ROOT
    from oracle import model as OM

    rng = np.random.default_rng(41)
    M = 400
    r = _fpn_like_rois(rng, M, 2, 1024)
    feats = [rng.standard_normal((2, 64, 256 >> i, 256 >> i)).astype(np.float32) for i in range(4)]
    per_image = [r[r[:, 0] == b][:, 1:] for b in (0, 1)]
    g = rng.standard_normal((M, 64, 7, 7)).astype(np.float32)
    src, dst = str(tmp_path / "in.npz"), str(tmp_path / "out.npz")
    np.savez(src, g=g, b0=per_image[0], b1=per_image[1], **{"f%d" % i: f for i, f in enumerate(feats)})
    env = dict(os.environ, JTSM_ALIGN_CENSUS_LIMIT="1")
    subprocess.run([sys.executable, "-c", _SCATTER_WORKER, ROOT, src, dst], env=env, check=True, timeout=600)
    got = np.load(dst)
    fr = [torch.from_numpy(f).requires_grad_() for f in feats]
    rois = torch.cat([torch.cat([torch.full((len(b), 1), float(i)), torch.from_numpy(b)], 1) for i, b in enumerate(per_image)])
    OM.roi_align_levels(fr, rois, 1024, 7).backward(torch.from_numpy(g))
    for i, b in enumerate(fr):
        ref = b.grad.numpy() if b.grad is not None else np.zeros_like(feats[i])     # (a level without rois)
        assert np.allclose(got["g%d" % i], ref, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(ref).max()))), i
    assert float(np.abs(got["g0"]).sum()) > 0


def test_fp16_boundary_moi_pool_exact_and_roi_align_rounded_once(cuda):
    """fp16 tensors at the pooling boundary (MOIPool_cuda.cu:400 dispatches on half; roi_align_rotated.py:79-85
    up-casts): MOIPool on fp16 features returns exactly the fp32 result on the same (fp16-representable) values, with
    the roi corners computed in half arithmetic; ROIAlign returns the fp32 result rounded once to fp16; backward too."""
    from jtsm_amd.layers.moi_pool import moi_pool
    from jtsm_amd.layers.roi_align import roi_align

    g = torch.Generator().manual_seed(5)
    B, C, H, W, L = 2, 64, 20, 24, 30
    x = torch.randn(B, C, H, W, generator=g).half()
    sp = (torch.arange(H * 4)[:, None] // 16 * 6 + torch.arange(W * 4)[None, :] // 16).to(torch.int32)[None].repeat(B, 1, 1)
    rois = torch.tensor([[0, 3.0, 2.0, 60.0, 50.0], [1, 10.5, 8.25, 70.0, 61.0], [0, 33.3, 20.7, 90.1, 77.9]])
    oh = (torch.rand(3, L, generator=g) < 0.7).to(torch.int32)
    xd = x.to(cuda).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out, arg = moi_pool(xd, rois.half().to(cuda), (7, 7), 0.25, oh.to(cuda), sp.to(cuda))
    assert out.dtype == torch.float16
    # fp32 run on the same values, corners as the half kernel computes them: round(Half(x) * Half(0.25)) in half
    pre = rois.clone()
    pre[:, 1:] = (rois[:, 1:].half().float() * 0.25).half().float()
    x32 = x.float().to(cuda).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out32, arg32 = moi_pool(x32, pre.to(cuda), (7, 7), 1.0, oh.to(cuda), sp.to(cuda))
    assert torch.equal(arg.cpu(), arg32.cpu()) and torch.equal(out.float().cpu(), out32.cpu())
    go = torch.randn(out.shape, generator=g).half()
    out.backward(go.to(cuda))
    out32.backward(go.float().to(cuda))
    assert torch.equal(xd.grad.float().cpu(), x32.grad.half().float().cpu())
    # ROIAlign: fp32 accumulate, one rounding
    y = roi_align(xd.detach(), rois.half().to(cuda), (7, 7), 0.25, 0, True)
    y32 = roi_align(x32.detach(), rois.half().float().to(cuda), (7, 7), 0.25, 0, True)
    assert y.dtype == torch.float16 and torch.equal(y.cpu(), y32.half().cpu())
