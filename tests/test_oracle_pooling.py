"""CPU suite: pins the ORACLE (oracle/c/pool_ops.inc) before anything trusts it.

  1. against the known answers the reference's own tests hold (kat_reference_tests.npz),
  2. against committed vectors produced by the reference's compiled CPU code (bit-exact),
  3. against that compiled code itself on fresh random inputs, when oracle/_ref is present,
  4. self-consistency: analytic backward vs numerical differentiation in fp64 (the reference's
     gradcheck recipe, tests/layers/test_roi_align_rotated.py:107-125), rotated(0 deg) == aligned
     (tests/modeling/test_roi_pooler.py:15-60), empty inputs.
MOIPool has no runnable reference: its vectors are regression pins ("parity unpinned").
"""
import numpy as np
import pytest

from conftest import load_cases
from oracle import pooling as P

KAT = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "kat_reference_tests.npz"))


def _simple(img, box, res, aligned):
    rois = np.array([[0, *box]], np.float32)
    return P.roi_align_forward(img[None, None].astype(np.float32), rois, 1.0, res[0], res[1], 0,
                               aligned)[0, 0]


def test_kat_forward_legacy_and_aligned():
    img = KAT["image5x5"]
    assert np.allclose(_simple(img, KAT["box"], (4, 4), False), KAT["legacy_4x4"])
    assert np.allclose(_simple(img, KAT["box"], (4, 4), True), KAT["aligned_4x4"])


def _rot90(a, k):
    for _ in range(k % 4):
        a = a.T[::-1]
    return a


def test_kat_rotated_0_90_180_270():
    img = KAT["image5x5"][None, None]
    b = KAT["box"]
    for i in range(4):
        roi = np.array([[0, (b[0] + b[2]) / 2, (b[1] + b[3]) / 2, b[2] - b[0], b[3] - b[1], 90 * i]],
                       np.float32)
        y = P.roi_align_rotated_forward(img, roi, 1.0, 4, 4, 0)[0, 0]
        assert np.allclose(y, _rot90(KAT["aligned_4x4"], -i)), i


def test_kat_empty_boxes_give_zero_output_and_zero_grad():
    img = np.random.default_rng(0).random((1, 1, 5, 5)).astype(np.float32)
    roi = np.array([[0, *KAT["empty_box"]]], np.float32)
    y = P.roi_align_forward(img, roi, 1.0, 7, 7, 0, True)
    assert y.shape == (1, 1, 7, 7) and (y == 0).all()
    gx = P.roi_align_backward(np.ones_like(y), roi, 1.0, 7, 7, 1, 1, 5, 5, 0, True)
    assert (gx == 0).all()
    rroi = np.array([[0, *KAT["rotated_empty_box"]]], np.float32)
    assert (P.roi_align_rotated_forward(img, rroi, 1.0, 7, 7, 0) == 0).all()


def test_empty_batch():
    y = P.roi_align_forward(np.zeros((0, 3, 10, 10), np.float32), np.zeros((0, 5), np.float32), 1.0,
                            7, 7, 0, True)
    assert y.shape == (0, 3, 7, 7)


def test_negative_extent_raises_like_reference_cpu():
    img = np.zeros((1, 1, 5, 5), np.float32)
    with pytest.raises(RuntimeError):
        P.roi_align_forward(img, np.array([[0, 4, 1, 2, 3]], np.float32), 1.0, 2, 2, 0, True)


@pytest.mark.parametrize("case", sorted(load_cases("roi_align_ref.npz")))
def test_roi_align_matches_reference_vectors_bit_exact(case):
    c = load_cases("roi_align_ref.npz")[case]
    scale, PH, PW, sr, al = c["meta"]
    PH, PW, sr, al = int(PH), int(PW), int(sr), bool(al)
    y = P.roi_align_forward(c["x"], c["rois"], scale, PH, PW, sr, al)
    assert np.array_equal(y, c["y"])
    B, Cc, H, W = c["x"].shape
    gx = P.roi_align_backward(c["g"], c["rois"], scale, PH, PW, B, Cc, H, W, sr, al)
    assert np.array_equal(gx, c["gx"])


@pytest.mark.parametrize("case", sorted(load_cases("roi_align_rotated_ref.npz")))
def test_roi_align_rotated_matches_reference_vectors_bit_exact(case):
    c = load_cases("roi_align_rotated_ref.npz")[case]
    scale, PH, PW, sr = c["meta"]
    PH, PW, sr = int(PH), int(PW), int(sr)
    y = P.roi_align_rotated_forward(c["x"], c["rois"], scale, PH, PW, sr)
    assert np.array_equal(y, c["y"])
    B, Cc, H, W = c["x"].shape
    gx = P.roi_align_rotated_backward(c["g"], c["rois"], scale, PH, PW, B, Cc, H, W, sr)
    assert np.array_equal(gx, c["gx"])


def test_against_compiled_reference_on_fresh_inputs(reference_module):
    if reference_module is None:
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    import torch

    ref, T = reference_module, torch.from_numpy
    rng = np.random.default_rng(7)
    for dt in (np.float32, np.float64):
        for _ in range(6):
            B, Cc, H, W, M = 2, 3, int(rng.integers(5, 40)), int(rng.integers(5, 40)), 19
            x = rng.standard_normal((B, Cc, H, W)).astype(dt)
            x0, y0 = rng.uniform(-5, W * 4, M), rng.uniform(-5, H * 4, M)
            w, h = rng.uniform(0, W * 4, M), rng.uniform(0, H * 4, M)
            b = rng.integers(0, B, M)
            r = np.stack([b, x0, y0, x0 + w, y0 + h], 1).astype(dt)
            rr = np.stack([b, x0 + w / 2, y0 + h / 2, w, h, rng.uniform(-180, 180, M)], 1).astype(dt)
            for sr in (0, 2):
                for al in (True, False):
                    assert np.array_equal(
                        P.roi_align_forward(x, r, 0.25, 7, 7, sr, al),
                        ref.roi_align_forward(T(x), T(r), 0.25, 7, 7, sr, al).numpy())
                y = P.roi_align_rotated_forward(x, rr, 0.25, 7, 7, sr)
                assert np.array_equal(y, ref.roi_align_rotated_forward(T(x), T(rr), 0.25, 7, 7, sr).numpy())
                g = rng.standard_normal(y.shape).astype(dt)
                assert np.array_equal(
                    P.roi_align_rotated_backward(g, rr, 0.25, 7, 7, B, Cc, H, W, sr),
                    ref.roi_align_rotated_backward(T(g), T(rr), 0.25, 7, 7, B, Cc, H, W, sr).numpy())
                assert np.array_equal(
                    P.roi_align_backward(g, r, 0.25, 7, 7, B, Cc, H, W, sr, True),
                    ref.roi_align_backward(T(g), T(r), 0.25, 7, 7, B, Cc, H, W, sr, True).numpy())


def test_strided_grad_is_honoured():
    rng = np.random.default_rng(3)
    rois = np.array([[0, 1, 1, 9, 7], [0, 2, 3, 6, 8]], np.float64)
    g = rng.standard_normal((2, 3, 5, 4))
    gt = np.ascontiguousarray(g.transpose(0, 1, 3, 2)).transpose(0, 1, 3, 2)  # same values, strided
    assert not gt.flags["C_CONTIGUOUS"]
    a = P.roi_align_backward(g, rois, 1.0, 5, 4, 1, 3, 10, 10, 2, True)
    b = P.roi_align_backward(gt, rois, 1.0, 5, 4, 1, 3, 10, 10, 2, True)
    assert np.array_equal(a, b)


def _numgrad(f, x, g, eps=1e-6):
    out = np.zeros_like(x)
    it = np.nditer(x, flags=["multi_index"])
    for _ in it:
        i = it.multi_index
        x[i] += eps
        hi = (f(x) * g).sum()
        x[i] -= 2 * eps
        lo = (f(x) * g).sum()
        x[i] += eps
        out[i] = (hi - lo) / (2 * eps)
    return out


def test_gradcheck_fp64_rotated_and_aligned():
    rng = np.random.default_rng(11)
    x = rng.random((1, 1, 10, 10))
    rr = KAT["grad_rois_rotated"]
    g = rng.standard_normal((3, 1, 5, 5))
    ana = P.roi_align_rotated_backward(g, rr, 0.5, 5, 5, 1, 1, 10, 10, 1)
    num = _numgrad(lambda v: P.roi_align_rotated_forward(v, rr, 0.5, 5, 5, 1), x.copy(), g)
    assert np.allclose(ana, num, atol=1e-6)
    ra = KAT["grad_rois_aligned"]
    ana = P.roi_align_backward(g, ra, 1.0, 5, 5, 1, 1, 10, 10, 2, True)
    num = _numgrad(lambda v: P.roi_align_forward(v, ra, 1.0, 5, 5, 2, True), x.copy(), g)
    assert np.allclose(ana, num, atol=1e-6)


def test_rotated_zero_angle_equals_aligned_and_gradients_agree():
    # tests/layers/test_roi_align_rotated.py:127-172 and tests/modeling/test_roi_pooler.py:15-60
    rng = np.random.default_rng(5)
    x = rng.random((1, 1, 10, 10))
    y_r = P.roi_align_rotated_forward(x, KAT["grad_rois_rotated"], 1.0, 5, 5, 2)
    y_a = P.roi_align_forward(x, KAT["grad_rois_aligned"], 1.0, 5, 5, 2, True)
    assert np.allclose(y_r, y_a, atol=1e-12)
    g = np.ones_like(y_a)
    assert np.allclose(
        P.roi_align_rotated_backward(g, KAT["grad_rois_rotated"], 1.0, 5, 5, 1, 1, 10, 10, 2),
        P.roi_align_backward(g, KAT["grad_rois_aligned"], 1.0, 5, 5, 1, 1, 10, 10, 2, True))


def test_sample_table_matches_forward():
    rng = np.random.default_rng(9)
    x = rng.standard_normal((1, 1, 12, 17)).astype(np.float32)
    roi = np.array([0, 3.3, 1.2, 40.7, 30.1], np.float32)
    grid, pos, w = P.roi_sample_table(roi, False, 12, 17, 0.25, 7, 7, 0, True)
    assert pos.shape[0] == grid[0] * grid[1] * 49
    plane = x[0, 0].ravel()
    v = (w * np.where(pos >= 0, plane[np.maximum(pos, 0)], 0)).reshape(49, -1, 4)
    y = P.roi_align_forward(x, roi[None], 0.25, 7, 7, 0, True)[0, 0].ravel()
    assert np.allclose(v.sum((1, 2)) / max(grid[0] * grid[1], 1), y, atol=1e-5)


# ------------------------------------------------------------------ MOIPool (unpinned) -----
@pytest.mark.parametrize("case", sorted(load_cases("moi_pool_oracle.npz")))
def test_moi_pool_regression_vectors(case):
    c = load_cases("moi_pool_oracle.npz")[case]
    scale, PH, PW = c["meta"]
    PH, PW = int(PH), int(PW)
    y, a = P.moi_pool_forward(c["x"], c["rois"], scale, PH, PW, c["oh"], c["sp"])
    assert np.array_equal(y, c["y"]) and np.array_equal(a, c["argmax"])
    B, Cc, H, W = c["x"].shape
    gx = P.moi_pool_backward(c["g"], c["rois"], a, scale, PH, PW, B, Cc, H, W)
    assert np.array_equal(gx, c["gx"])


def _moi_bruteforce(x, rois, scale, PH, PW, oh, sp):
    """Pure-Python reading of SURVEY Appendix A.3 (small cases only)."""
    import math

    B, Cc, H, W = x.shape
    Hs, Ws = sp.shape[1:]
    M = rois.shape[0]
    out = np.zeros((M, Cc, PH, PW), np.float32)
    arg = -np.ones((M, Cc, PH, PW), np.int32)
    rnd = lambda v: int(math.floor(abs(v) + 0.5) * (1 if v >= 0 else -1))  # half away from zero
    s = np.float32(H / Hs)
    for n in range(M):
        b = int(rois[n, 0])
        x0, y0, x1, y1 = [rnd(float(np.float32(rois[n, i]) * np.float32(scale))) for i in (1, 2, 3, 4)]
        rw, rh = max(x1 - x0 + 1, 1), max(y1 - y0 + 1, 1)
        bh, bw = np.float32(rh) / np.float32(PH), np.float32(rw) / np.float32(PW)
        moi = np.zeros((H, W), bool)
        for h in range(max(y0, 0), min(y1, H - 1) + 1):
            for w in range(max(x0, 0), min(x1, W - 1) + 1):
                hs, he = int(np.floor(np.float32(h) / s)), int(np.ceil(np.float32(h + 1) / s))
                ws, we = int(np.floor(np.float32(w) / s)), int(np.ceil(np.float32(w + 1) / s))
                ids = sp[b, max(hs, 0):min(he, Hs), max(ws, 0):min(we, Ws)]
                moi[h, w] = bool((oh[n, ids.ravel()] == 1).any())
        for ph in range(PH):
            for pw in range(PW):
                hs = min(max(int(np.floor(np.float32(ph) * bh)) + y0, 0), H)
                he = min(max(int(np.ceil(np.float32(ph + 1) * bh)) + y0, 0), H)
                ws = min(max(int(np.floor(np.float32(pw) * bw)) + x0, 0), W)
                we = min(max(int(np.ceil(np.float32(pw + 1) * bw)) + x0, 0), W)
                for c in range(Cc):
                    best, at = -np.inf, -1
                    for h in range(hs, he):
                        for w in range(ws, we):
                            if moi[h, w] and x[b, c, h, w] > best:
                                best, at = x[b, c, h, w], h * W + w
                    if at >= 0:
                        out[n, c, ph, pw], arg[n, c, ph, pw] = best, at
    return out, arg


def test_moi_pool_against_python_reading_of_spec():
    rng = np.random.default_rng(21)
    B, Cc, H, W, stride = 2, 2, 6, 7, 4
    Hs, Ws = H * stride, W * stride
    sp = rng.integers(0, 12, (B, Hs // 4, Ws // 4)).repeat(4, 1).repeat(4, 2).astype(np.int32)
    sp = np.roll(sp, 2, 2)
    x = (np.round(rng.standard_normal((B, Cc, H, W)) * 2) / 2).astype(np.float32)
    rois = np.array([[0, 1, 2, 20, 19], [1, 0, 0, 27, 23], [1, 9.5, 4.5, 10.4, 22], [0, 30, 30, 2, 2]],
                    np.float32)
    oh = (rng.uniform(size=(4, 12)) < 0.4).astype(np.int32)
    y, a = P.moi_pool_forward(x, rois, 0.25, 3, 3, oh, sp)
    yb, ab = _moi_bruteforce(x, rois, 0.25, 3, 3, oh, sp)
    assert np.array_equal(a, ab) and np.array_equal(y, yb)
    mois = P.moi_mask(rois, oh, sp, H, W, 0.25)
    assert mois.shape == (4, H, W) and set(np.unique(mois)) <= {0, 1}
    # a bin is non-empty iff some masked cell lies in it
    assert ((a[:, 0] >= 0).any((1, 2)) == (mois.reshape(4, -1).sum(1) > 0)).all()
