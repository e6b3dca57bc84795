"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes front end of oracle/liboracle.so (oracle/c/pool_ops.inc): numpy in, numpy out.
Signatures mirror the reference FFI they restate:
  roi_align_forward/backward          detectron2/layers/csrc/ROIAlign/ROIAlign.h:7-27
  roi_align_rotated_forward/backward  detectron2/layers/csrc/ROIAlignRotated/ROIAlignRotated.h:7-27
  moi_pool_forward/backward           projects/WSL/wsl/layers/csrc/MOIPool/MOIPool.h:7-47
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.isfile(path):
            subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
        _LIB = C.CDLL(path)
    return _LIB


def _sfx(dt):
    dt = np.dtype(dt)
    if dt == np.float32:
        return "_f32", C.c_float
    if dt == np.float64:
        return "_f64", C.c_double
    raise TypeError("oracle supports float32/float64, got %s" % dt)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _elem_strides(a):
    return [C.c_long(s // a.itemsize) for s in a.strides]


def roi_align_forward(inp, rois, spatial_scale, pooled_h, pooled_w, sampling_ratio, aligned):
    pooled_h, pooled_w, sampling_ratio = int(pooled_h), int(pooled_w), int(sampling_ratio)
    inp = np.ascontiguousarray(inp)
    rois = np.ascontiguousarray(rois, dtype=inp.dtype)
    sfx, ct = _sfx(inp.dtype)
    B, Cc, H, W = inp.shape
    M = rois.shape[0]
    out = np.zeros((M, Cc, pooled_h, pooled_w), inp.dtype)
    if out.size == 0:
        return out
    rc = getattr(lib(), "oracle_roi_align_forward" + sfx)(
        _p(inp), _p(rois), _p(out), B, Cc, H, W, M, ct(spatial_scale), pooled_h, pooled_w,
        sampling_ratio, int(bool(aligned)))
    if rc:
        raise RuntimeError("ROIs in ROIAlign cannot have negative size")
    return out


def roi_align_backward(grad, rois, spatial_scale, pooled_h, pooled_w, B, Cc, H, W,
                       sampling_ratio, aligned):
    sfx, ct = _sfx(grad.dtype)
    rois = np.ascontiguousarray(rois, dtype=grad.dtype)
    B, Cc, H, W = int(B), int(Cc), int(H), int(W)
    gin = np.zeros((B, Cc, H, W), grad.dtype)
    if grad.size == 0:
        return gin
    rc = getattr(lib(), "oracle_roi_align_backward" + sfx)(
        _p(grad), _p(rois), _p(gin), B, Cc, H, W, rois.shape[0], ct(spatial_scale), pooled_h,
        pooled_w, sampling_ratio, int(bool(aligned)), *_elem_strides(grad))
    if rc:
        raise RuntimeError("ROIs in ROIAlign cannot have negative size")
    return gin


def roi_align_rotated_forward(inp, rois, spatial_scale, pooled_h, pooled_w, sampling_ratio):
    inp = np.ascontiguousarray(inp)
    rois = np.ascontiguousarray(rois, dtype=inp.dtype)
    sfx, ct = _sfx(inp.dtype)
    B, Cc, H, W = inp.shape
    M = rois.shape[0]
    out = np.zeros((M, Cc, pooled_h, pooled_w), inp.dtype)
    if out.size == 0:
        return out
    rc = getattr(lib(), "oracle_roi_align_rotated_forward" + sfx)(
        _p(inp), _p(rois), _p(out), B, Cc, H, W, M, ct(spatial_scale), pooled_h, pooled_w,
        sampling_ratio)
    if rc:
        raise RuntimeError("ROIs in ROIAlignRotated cannot have negative size")
    return out


def roi_align_rotated_backward(grad, rois, spatial_scale, pooled_h, pooled_w, B, Cc, H, W,
                               sampling_ratio):
    sfx, ct = _sfx(grad.dtype)
    rois = np.ascontiguousarray(rois, dtype=grad.dtype)
    B, Cc, H, W = int(B), int(Cc), int(H), int(W)
    gin = np.zeros((B, Cc, H, W), grad.dtype)
    if grad.size == 0:
        return gin
    rc = getattr(lib(), "oracle_roi_align_rotated_backward" + sfx)(
        _p(grad), _p(rois), _p(gin), B, Cc, H, W, rois.shape[0], ct(spatial_scale), pooled_h,
        pooled_w, sampling_ratio, *_elem_strides(grad))
    if rc:
        raise RuntimeError("ROIs in ROIAlignRotated cannot have negative size")
    return gin


def roi_sample_table(roi, rotated, H, W, spatial_scale, pooled_h, pooled_w, sampling_ratio,
                     aligned=True, cap=1 << 20):
    """(grid[2], pos[n,4] int32 (-1 = out of range), w[n,4]) of ONE roi: the bit-exact contract."""
    roi = np.ascontiguousarray(roi)
    sfx, ct = _sfx(roi.dtype)
    grid = np.zeros(2, np.int32)
    pos = np.zeros((cap, 4), np.int32)
    w = np.zeros((cap, 4), roi.dtype)
    n = getattr(lib(), "oracle_roi_sample_table" + sfx)(
        _p(roi), int(bool(rotated)), H, W, ct(spatial_scale), pooled_h, pooled_w, sampling_ratio,
        int(bool(aligned)), _p(grid), _p(pos), _p(w), cap)
    if n < 0:
        raise ValueError("sample table larger than cap")
    return grid, pos[:n].copy(), w[:n].copy()


def moi_mask(rois, oh_labels, superpixels, H, W, spatial_scale):
    rois = np.ascontiguousarray(rois)
    sfx, ct = _sfx(rois.dtype)
    oh = np.ascontiguousarray(oh_labels, dtype=np.int32)
    sp = np.ascontiguousarray(superpixels, dtype=np.int32)
    M = rois.shape[0]
    mois = np.zeros((M, H, W), np.int32)
    if mois.size:
        getattr(lib(), "oracle_moi_mask" + sfx)(
            _p(rois), _p(oh), _p(sp), _p(mois), M, oh.shape[1], H, W, sp.shape[1], sp.shape[2],
            ct(spatial_scale))
    return mois


def moi_pool_forward(inp, rois, spatial_scale, pooled_h, pooled_w, oh_labels, superpixels):
    inp = np.ascontiguousarray(inp)
    rois = np.ascontiguousarray(rois, dtype=inp.dtype)
    sfx, ct = _sfx(inp.dtype)
    oh = np.ascontiguousarray(oh_labels, dtype=np.int32)
    sp = np.ascontiguousarray(superpixels, dtype=np.int32)
    B, Cc, H, W = inp.shape
    M = rois.shape[0]
    out = np.zeros((M, Cc, pooled_h, pooled_w), inp.dtype)
    arg = np.zeros((M, Cc, pooled_h, pooled_w), np.int32)
    if out.size:
        getattr(lib(), "oracle_moi_pool_forward" + sfx)(
            _p(inp), _p(rois), _p(oh), _p(sp), _p(out), _p(arg), B, Cc, H, W, M, oh.shape[1],
            sp.shape[1], sp.shape[2], ct(spatial_scale), pooled_h, pooled_w)
    return out, arg


def moi_pool_backward(grad, rois, argmax, spatial_scale, pooled_h, pooled_w, B, Cc, H, W):
    sfx, _ = _sfx(grad.dtype)
    rois = np.ascontiguousarray(rois, dtype=grad.dtype)
    arg = np.ascontiguousarray(argmax, dtype=np.int32)
    B, Cc, H, W = int(B), int(Cc), int(H), int(W)
    gin = np.zeros((B, Cc, H, W), grad.dtype)
    if grad.size:
        getattr(lib(), "oracle_moi_pool_backward" + sfx)(
            _p(grad), _p(rois), _p(arg), _p(gin), B, Cc, H, W, rois.shape[0], pooled_h, pooled_w,
            *_elem_strides(grad))
    return gin
