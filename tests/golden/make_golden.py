"""Generate the committed golden fixtures under tests/golden/.

Run in the build container only (needs /root/reference for the pooling vectors):

    python oracle/build_ref.py && python tests/golden/make_golden.py

Sources of truth
  * roi_align_ref.npz / roi_align_rotated_ref.npz: outputs of the REFERENCE's own CPU code
    (detectron2/layers/csrc/ROIAlign/ROIAlign_cpu.cpp, .../ROIAlignRotated_cpu.cpp) compiled
    by oracle/build_ref.py, run on seeded inputs.  Data only: inputs + outputs.
  * kat_reference_tests.npz: the known-answer tables the reference's tests hold
    (tests/layers/test_roi_align.py:22-45, tests/layers/test_roi_align_rotated.py:30-71,
    gradient ROI set :107-125).
  * nms_ref.npz: keep lists of the REFERENCE's own greedy NMS loop (detectron2/layers/csrc/nms_rotated/
    nms_rotated_cpu.cpp — "modified from torchvision's nms_cpu_kernel", its comment says — compiled by
    oracle/build_ref.py) on axis-aligned boxes passed as (cx, cy, w, h, 0).  Its IoU comes from polygon clipping and
    its test is `>=`, so every case is generated with all pairwise IoUs at least 1e-3 away from the threshold: there
    the keep list depends only on the greedy visiting order and the suppress rule, which is what it pins.
  * moi_pool_oracle.npz: MOIPool has no runnable reference -> vectors come from
    our restatement (oracle/c/pool_ops.inc); "parity unpinned", they pin regressions only.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))

from oracle import pooling as P  # noqa: E402
from oracle.build_ref import build, load_prebuilt  # noqa: E402


def boxes(rng, M, B, Wimg, Himg, dtype):
    x0 = rng.uniform(-8, Wimg, M)
    y0 = rng.uniform(-8, Himg, M)
    w = np.exp(rng.uniform(np.log(2), np.log(Wimg), M))
    h = np.exp(rng.uniform(np.log(2), np.log(Himg), M))
    b = rng.integers(0, B, M)
    return np.stack([b, x0, y0, x0 + w, y0 + h], 1).astype(dtype)


def main():
    ref = load_prebuilt() or build()
    assert ref is not None, "reference build unavailable"
    rng = np.random.default_rng(20261003)
    T = torch.from_numpy

    # ---- ROIAlign: several (sampling_ratio, aligned, scale, pooled) cases -----------------
    cases = {}
    for name, (B, Cc, H, W, M, scale, PH, PW, sr, al, dt) in {
        "a_s0_al": (2, 8, 25, 31, 24, 0.25, 7, 7, 0, True, np.float32),
        "b_s2_al": (2, 4, 16, 16, 16, 0.125, 7, 7, 2, True, np.float32),
        "c_s0_legacy": (1, 4, 20, 14, 16, 0.5, 5, 3, 0, False, np.float32),
        "d_s0_al_f64": (2, 3, 12, 18, 12, 0.25, 14, 14, 0, True, np.float64),
    }.items():
        x = rng.standard_normal((B, Cc, H, W)).astype(dt)
        r = boxes(rng, M, B, W / scale, H / scale, dt)
        r[0, 1:] = [3, 4, 5, 4] if al else r[0, 1:]  # zero-height box (reference test_empty_box)
        y = ref.roi_align_forward(T(x), T(r), scale, PH, PW, sr, al).numpy()
        g = rng.standard_normal(y.shape).astype(dt)
        gx = ref.roi_align_backward(T(g), T(r), scale, PH, PW, B, Cc, H, W, sr, al).numpy()
        cases[name] = dict(x=x, rois=r, y=y, g=g, gx=gx,
                           meta=np.array([scale, PH, PW, sr, int(al)], np.float64))
    np.savez_compressed(os.path.join(HERE, "roi_align_ref.npz"),
                        **{"%s__%s" % (k, f): v for k, d in cases.items() for f, v in d.items()})

    # ---- ROIAlignRotated ---------------------------------------------------------------
    cases = {}
    for name, (B, Cc, H, W, M, scale, PH, PW, sr, dt) in {
        "a_s0": (2, 8, 25, 31, 24, 0.25, 7, 7, 0, np.float32),
        "b_s2": (1, 4, 16, 16, 16, 0.5, 5, 5, 2, np.float32),
        "c_s0_f64": (2, 3, 12, 18, 12, 0.25, 7, 7, 0, np.float64),
    }.items():
        x = rng.standard_normal((B, Cc, H, W)).astype(dt)
        bx = boxes(rng, M, B, W / scale, H / scale, np.float64)
        ang = rng.uniform(-180, 180, M)
        ang[:4] = [0, 90, 180, 270]
        r = np.stack([bx[:, 0], (bx[:, 1] + bx[:, 3]) / 2, (bx[:, 2] + bx[:, 4]) / 2,
                      bx[:, 3] - bx[:, 1], bx[:, 4] - bx[:, 2], ang], 1).astype(dt)
        y = ref.roi_align_rotated_forward(T(x), T(r), scale, PH, PW, sr).numpy()
        g = rng.standard_normal(y.shape).astype(dt)
        gx = ref.roi_align_rotated_backward(T(g), T(r), scale, PH, PW, B, Cc, H, W, sr).numpy()
        cases[name] = dict(x=x, rois=r, y=y, g=g, gx=gx,
                           meta=np.array([scale, PH, PW, sr], np.float64))
    np.savez_compressed(os.path.join(HERE, "roi_align_rotated_ref.npz"),
                        **{"%s__%s" % (k, f): v for k, d in cases.items() for f, v in d.items()})

    # ---- known answers held by the reference's tests (data) ---------------------------------
    np.savez_compressed(
        os.path.join(HERE, "kat_reference_tests.npz"),
        image5x5=np.arange(25, dtype=np.float32).reshape(5, 5),
        box=np.array([1, 1, 3, 3], np.float32),
        legacy_4x4=np.array([[7.5, 8, 8.5, 9], [10, 10.5, 11, 11.5], [12.5, 13, 13.5, 14],
                             [15, 15.5, 16, 16.5]], np.float32),
        aligned_4x4=np.array([[4.5, 5.0, 5.5, 6.0], [7.0, 7.5, 8.0, 8.5], [9.5, 10.0, 10.5, 11.0],
                              [12.0, 12.5, 13.0, 13.5]], np.float32),
        empty_box=np.array([3, 4, 5, 4], np.float32),
        rotated_empty_box=np.array([2, 3, 0, 0, 0], np.float32),
        grad_rois_rotated=np.array([[0, 4.5, 4.5, 9, 9, 0], [0, 2, 7, 4, 4, 0], [0, 7, 7, 4, 4, 0]],
                                   np.float64),
        grad_rois_aligned=np.array([[0, 0, 0, 9, 9], [0, 0, 5, 4, 9], [0, 5, 5, 9, 9]], np.float64),
    )

    # ---- MOIPool (oracle-generated; parity unpinned) ------------------------------------------
    cases = {}
    for name, (B, Cc, H, W, M, stride, PH, PW, blk) in {
        "a_stride4": (2, 8, 24, 32, 24, 4, 7, 7, 16),
        "b_stride16": (2, 4, 8, 10, 16, 16, 7, 7, 32),
        "c_stride8_3x5": (1, 4, 16, 12, 12, 8, 3, 5, 24),
    }.items():
        Hs, Ws = H * stride, W * stride
        gy, gx_ = (Hs + blk - 1) // blk, (Ws + blk - 1) // blk
        Lw = gy * gx_
        ids = (np.arange(Hs)[:, None] // blk) * gx_ + (np.arange(Ws)[None, :] // blk)
        # jitter block borders so cells straddle several superpixels
        sp = np.stack([np.roll(ids, (int(rng.integers(0, blk)), int(rng.integers(0, blk))), (0, 1))
                       for _ in range(B)]).astype(np.int32)
        x = rng.standard_normal((B, Cc, H, W)).astype(np.float32)
        x = np.round(x * 4) / 4  # coarse values -> many exact ties inside a bin
        r = boxes(rng, M, B, Ws, Hs, np.float32)
        r[1, 1:] = [40, 40, 30, 30]  # malformed (x1 < x0): forced to 1x1, every cell fails inside test
        oh = (rng.uniform(size=(M, Lw)) < 0.35).astype(np.int32)
        oh[2] = 0  # roi without any labelled superpixel -> all bins empty
        oh[3] = 2  # values other than 1 do not count
        y, a = P.moi_pool_forward(x, r, 1.0 / stride, PH, PW, oh, sp)
        g = rng.standard_normal(y.shape).astype(np.float32)
        gx = P.moi_pool_backward(g, r, a, 1.0 / stride, PH, PW, B, Cc, H, W)
        cases[name] = dict(x=x, rois=r, oh=oh, sp=sp, y=y, argmax=a, g=g, gx=gx,
                           meta=np.array([1.0 / stride, PH, PW], np.float64))
    np.savez_compressed(os.path.join(HERE, "moi_pool_oracle.npz"),
                        **{"%s__%s" % (k, f): v for k, d in cases.items() for f, v in d.items()})
    # ---- greedy NMS: the reference's own loop on angle-0 boxes -------------------------------------
    cases = {}
    for name, (n, size, thr) in {"n200_t03": (200, 120.0, 0.3), "n400_t05": (400, 200.0, 0.5),
                                 "n300_t07": (300, 150.0, 0.7), "n64_t05": (64, 40.0, 0.5)}.items():
        for attempt in range(200):
            xy = rng.uniform(0, size * 0.8, (n, 2))
            wh = rng.uniform(2.0, size * 0.3, (n, 2))
            b = np.concatenate([xy, xy + wh], 1).astype(np.float32)
            sc = rng.permutation(n).astype(np.float32) / n          # distinct scores: no ordering ambiguity
            bd = b.astype(np.float64)
            iw = np.clip(np.minimum(bd[:, None, 2], bd[None, :, 2]) - np.maximum(bd[:, None, 0], bd[None, :, 0]), 0, None)
            ih = np.clip(np.minimum(bd[:, None, 3], bd[None, :, 3]) - np.maximum(bd[:, None, 1], bd[None, :, 1]), 0, None)
            inter = iw * ih
            area = (bd[:, 2] - bd[:, 0]) * (bd[:, 3] - bd[:, 1])
            iou = inter / (area[:, None] + area[None, :] - inter)
            np.fill_diagonal(iou, 0.0)
            if np.abs(iou - thr).min() > 1e-3:
                break
        else:
            raise RuntimeError("no NMS case away from the threshold found")
        rb = np.stack([(b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2, b[:, 2] - b[:, 0], b[:, 3] - b[:, 1],
                       np.zeros(n, np.float32)], 1).astype(np.float32)
        keep = ref.nms_rotated(T(rb), T(sc), float(thr)).numpy()
        cases[name] = dict(boxes=b, scores=sc, keep=keep.astype(np.int64), thr=np.array([thr], np.float64))
    np.savez_compressed(os.path.join(HERE, "nms_ref.npz"),
                        **{"%s__%s" % (k, f): v for k, d in cases.items() for f, v in d.items()})
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
