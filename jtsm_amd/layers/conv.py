"""Convolution / linear primitives of the hot path, on libjtsm_hip.so's fp32-MFMA implicit GEMM.

Tensors keep the reference's logical shapes — activations (N,C,H,W), weights (O,I,kh,kw) — but
must be stored channels_last (NHWC / OHWI in memory), which is what the MI355X kernels read.
`conv2d_fused` is the autograd-aware functional form of what the reference runs as
Conv2d -> FrozenBatchNorm2d -> (+shortcut) -> ReLU (detectron2/layers/wrappers.py:62-83,
batch_norm.py:45-66, modeling/backbone/resnet.py:195-211).
"""
import ctypes as C
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib as L
from . import grad_fan

CL = torch.channels_last

# Optional per-launch timing (bench.py's roofline leg): when a list, every contraction launch appends
# (kernel variant, algorithmic FLOPs, LaunchSpan, shape); events are recorded on the launch stream.
LAUNCH_LOG = None
# Which part of the model the next contraction launches belong to ("backbone" | "fpn" | "heads"): set by the modules on
# their way forward, remembered by every autograd node of this package and restored at the start of its backward.  Only
# the measurement tools read it (LAUNCH_LOG entries carry it: bench.py --launch-sequence, tools/pmc_mfma.py).
SEGMENT = "heads"


def set_segment(name):
    global SEGMENT
    SEGMENT = name


# Contraction arithmetic: "f32" = exact fp32 MFMA; "bf16x3" = split-bf16 (three bf16 MFMA products per fp32
# product, fp32 accumulate, ~2^-16 relative error per product) for every eligible forward / data-gradient
# contraction; "f16" = ONE fp16 plane per operand, one fp16 MFMA product, fp32 accumulate and fp32 results
# (BASELINE configs[4]; ~2^-11 relative error per operand — NOT inside the 1e-4 parity bar, never the headline).
# See csrc/conv_x3.h.
MATH = os.environ.get("JTSM_CONV_MATH", "bf16x3")
_MODES = ("f32", "bf16x3", "f16")
# fp16 mode: gradient planes are stored times 2^GRAD_SHIFT (exact) so small gradients stay clear of fp16's
# subnormal range; the kernels undo it in the accumulator (include/jtsm_hip.h, "fp16 contractions")
GRAD_SHIFT = int(os.environ.get("JTSM_F16_GRAD_SHIFT", "12"))
# bias gradients inside the weight-gradient contraction (csrc/conv_x3.h: x3_bias_mma); 0: separate channel-sum passes
# (the A/B switch of tools/sweeps)
BIAS_IN_WGRAD = os.environ.get("JTSM_BIAS_IN_WGRAD", "1") != "0"


def planes_mode():
    """True when the contractions read operand planes (bf16 hi/lo pairs or one fp16 plane)."""
    return MATH != "f32"


def set_math(mode):
    global MATH
    if mode not in _MODES:
        raise ValueError("conv math must be one of %r, got %r" % (_MODES, mode))
    if mode != MATH:   # cached planes are in the old mode's format
        _drop_planes()
    MATH = mode


def _planes_buf(n, device):
    n8 = max((n + 7) // 8 * 8, 64)   # (separate planes stay more than 32 elements apart: lo == hi + 32 means PAIRED)
    if MATH == "f16":
        return torch.empty(n8, dtype=torch.int16, device=device)   # one fp16 plane
    return torch.empty(2 * n8, dtype=torch.int16, device=device)   # hi plane, then lo plane (16-byte aligned)


def _hl(buf):
    """(hi, lo) device pointers of a planes buffer (lo is None for the single fp16 plane).  A PAIRED buffer (weights,
    `_paired_ok`: per row, blocks of 32 hi values followed by their 32 lo values) is named by lo == hi + 64 bytes."""
    if buf is None:
        return None, None
    p = buf.data_ptr()
    if MATH == "f16":
        return C.c_void_p(p), None
    if getattr(buf, "_paired", False):
        return C.c_void_p(p), C.c_void_p(p + 64)
    return C.c_void_p(p), C.c_void_p(p + buf.numel())   # lo starts numel/2 int16 = numel bytes in


# Weights as PAIRED planes (csrc/conv_x3.h `x3_paired`): one K stage of a contraction is one whole 128-byte line per
# weight row instead of half a line from each of two arrays.  JTSM_W_PAIRED=0 keeps separate planes (sweeps).
W_PAIRED = os.environ.get("JTSM_W_PAIRED", "1") != "0"


def _paired_ok(w, transposed):
    """Row length of the weight operand ([out][taps*in], transposed: [in][taps*out]) when it can be paired, else 0."""
    if not W_PAIRED or MATH != "bf16x3" or w.numel() == 0:
        return 0
    k = w.numel() // w.shape[1 if transposed else 0]
    return k if k % 32 == 0 else 0


def _split(t, grad=False):
    flat = t.permute(0, 2, 3, 1) if t.dim() == 4 else t
    if not flat.is_contiguous():
        raise RuntimeError("split_bf16: tensor must be channels_last (4-d) or contiguous")
    buf = _planes_buf(t.numel(), t.device)
    hi, lo = _hl(buf)
    L.note_bytes((6.0 if MATH == "f16" else 8.0) * t.numel())   # fp32 read, planes written
    if MATH == "f16":
        L.check(L.lib().jtsm_split_f16_f32(L.ptr(t), hi, C.c_long(t.numel()), GRAD_SHIFT if grad else 0, L.stream()),
                "split_f16")
        return buf
    L.check(L.lib().jtsm_split_bf16_f32(L.ptr(t), hi, lo, C.c_long(t.numel()), L.stream()), "split_bf16")
    return buf


def split_bf16(t):
    """(hi, lo) bf16 planes (int16 bit patterns) of a float32 tensor, in its storage order."""
    buf = _split(t)
    n8 = buf.numel() // 2
    return buf[:t.numel()], buf[n8:n8 + t.numel()]


def _split_paired(w):
    """Paired planes of a weight [out][taps][in] in its storage order (rows of k = taps * in)."""
    k = _paired_ok(w, False)
    buf = _planes_buf(w.numel(), w.device)
    buf._paired = True
    L.note_bytes(8.0 * w.numel())
    L.check(L.lib().jtsm_split_bf16_paired_f32(L.ptr(w), C.c_void_p(buf.data_ptr()), C.c_long(w.numel() // k), k,
                                               L.stream()), "split_bf16_paired")
    return buf


def _split_transposed_into(buf, w, row_scale, paired=True):
    o, i, kh, kw = w.shape
    if paired and _paired_ok(w, True):
        buf._paired = True
    hi, lo = _hl(buf)
    if MATH == "f16":
        L.check(L.lib().jtsm_split_f16_transposed_f32(L.ptr(w), L.ptr(row_scale), hi, o, kh * kw, i, L.stream()),
                "split_f16_transposed")
    else:
        L.check(L.lib().jtsm_split_bf16_transposed_f32(L.ptr(w), L.ptr(row_scale), hi, lo, o, kh * kw, i, L.stream()),
                "split_bf16_transposed")
    return buf


def _split_transposed(w, row_scale=None, paired=True):
    return _split_transposed_into(_planes_buf(w.numel(), w.device), w, row_scale, paired)


def split_bf16_transposed(w, row_scale=None):
    """Planes of W^T [in][taps][out] of a channels_last (out, in, kh, kw) weight, rows pre-multiplied by
    row_scale[out] when given."""
    buf = _split_transposed(w, row_scale, paired=False)   # (two separate planes: what this helper's callers index)
    n8 = buf.numel() // 2
    return buf[:w.numel()], buf[n8:n8 + w.numel()]


# bf16 planes already made this step, keyed by the tensor's memory: a conv epilogue or relu_backward that
# emitted the planes of its output registers them here, and the next contraction that consumes the tensor
# (forward input, weight-gradient input, shared block input of conv1 + shortcut + FPN lateral) finds them.
# Entries hold the tensor (detached), so its memory cannot be recycled under a live key; the model clears
# the cache at the start of every forward (planes_clear).
_PLANES = {}
_PLANES_MAX_BYTES = 16 << 30   # entries pin their tensors: a caller that never starts a new step is bounded here
_planes_bytes = 0


# callables run by planes_clear(): per-step state other modules key on this cache's lifetime (layers/fused_blocks.py
# drops its held block-output gradient)
CLEAR_HOOKS = []


def planes_clear():
    global _planes_bytes
    _PLANES.clear()
    _planes_bytes = 0
    for hook in CLEAR_HOOKS:
        hook()
    refresh_weight_planes()


def _drop_planes():
    """Forget every cached plane (activations and weights): the arithmetic mode is changing."""
    global _planes_bytes
    _PLANES.clear()
    _planes_bytes = 0
    _WPLANES.clear()
    _WTABLES.clear()


def planes_put(t, buf):
    global _planes_bytes
    if _planes_bytes > _PLANES_MAX_BYTES:
        _PLANES.clear()
        _planes_bytes = 0
    # planes mirror the flat memory of a DENSE tensor, so any dense view of the same bytes shares them
    _PLANES[(t.data_ptr(), t.numel())] = (t.detach(), t._version, buf)
    _planes_bytes += 8 * t.numel()


def planes_forget(t):
    """A kernel is about to add into t through its raw pointer (no version bump): cached planes of it are stale."""
    _PLANES.pop((t.data_ptr(), t.numel()), None)


def planes_of(t, grad=False):
    """Cached planes buffer of a tensor, splitting it now if nobody has.  grad: the tensor is a gradient (fp16 mode
    stores those planes times 2^GRAD_SHIFT; whoever registered planes for a gradient applied the same factor)."""
    e = _PLANES.get((t.data_ptr(), t.numel()))
    if e is not None and e[1] == t._version:
        return e[2]
    buf = _split(t, grad)
    planes_put(t, buf)
    return buf


# Weight planes live across steps: a weight is re-split only when its version counter has moved (the
# optimizer's in-place update), frozen weights never again.  At the start of a step (planes_clear) every stale
# entry is refreshed by ONE launch per form (straight / transposed+scaled) instead of ~150 small ones.
class _WEntry(object):
    __slots__ = ("w", "scale", "transposed", "buf", "version")


_WPLANES = {}
_WTABLES = {}   # (transposed, tuple of entry keys) -> (device table, blocks): rebuilt only when the stale set changes


def _cacheable_weight(w):
    """Planes persist across steps only for real parameters (or views of one): under no_grad every temporary is a
    leaf too, and caching those (concatenated / zero-padded / reshaped weights) would pin a fresh entry per call."""
    base = w._base if w._base is not None else w
    return isinstance(base, torch.nn.Parameter) and (w.dim() != 4 or w.permute(0, 2, 3, 1).is_contiguous())


def _weight_planes(w, transposed=False, scale=None):
    """Planes of a weight ([out][taps][in], or transposed+row-scaled for the data gradient), cached by version."""
    if not _cacheable_weight(w):
        if transposed:
            return _split_transposed(w, scale)
        return _split_paired(w) if _paired_ok(w, False) else _split(w)
    key = (w.data_ptr(), w.numel(), transposed, scale.data_ptr() if scale is not None else 0)
    e = _WPLANES.get(key)
    if e is not None and e.version == w._version:
        return e.buf
    if e is None:
        if len(_WPLANES) >= 2048:   # (entries pin their weights; a process that keeps building models starts over)
            _WPLANES.clear()
            _WTABLES.clear()
        e = _WEntry()
        e.w, e.scale, e.transposed = w.detach(), scale, transposed
        e.buf = _planes_buf(w.numel(), w.device)
        e.buf._paired = bool(_paired_ok(w, transposed))
        _WPLANES[key] = e
    hi, lo = _hl(e.buf)
    lib = L.lib()
    if transposed:
        _split_transposed_into(e.buf, w, scale)
    elif e.buf._paired:
        L.check(lib.jtsm_split_bf16_paired_f32(L.ptr(w), hi, C.c_long(w.shape[0]), w.numel() // w.shape[0], L.stream()),
                "split_bf16_paired")
    elif MATH == "f16":
        L.check(lib.jtsm_split_f16_f32(L.ptr(w), hi, C.c_long(w.numel()), 0, L.stream()), "split_f16")
    else:
        L.check(lib.jtsm_split_bf16_f32(L.ptr(w), hi, lo, C.c_long(w.numel()), L.stream()), "split_bf16")
    e.version = w._version
    return e.buf


def refresh_weight_planes():
    """Re-split every cached weight whose version moved, one launch per form."""
    if MATH == "f32" or not _WPLANES:
        return
    for transposed in (False, True):
        stale = [(k, e) for k, e in _WPLANES.items() if e.transposed == transposed and e.version != e.w._version]
        if not stale:
            continue
        tkey = (transposed, tuple(k for k, _ in stale))
        tab = _WTABLES.get(tkey)
        if tab is None:
            rows, blocks = [], 0
            for _, e in stale:
                hi = e.buf.data_ptr()
                lo = 0 if MATH == "f16" else hi + e.buf.numel()   # lo == 0: the record's fp16 plane goes to hi
                paired = getattr(e.buf, "_paired", False)
                if paired:
                    lo = hi + 64
                if transposed:
                    o, i, kh, kw = e.w.shape
                    nb = ((i + 31) // 32) * ((o + 31) // 32) * kh * kw
                    rows.append([e.w.data_ptr(), hi, lo, e.scale.data_ptr() if e.scale is not None else 0, blocks, o,
                                 kh * kw, i])
                else:
                    nb = (e.w.numel() + 2047) // 2048
                    rows.append([e.w.data_ptr(), hi, lo, 0, blocks, e.w.numel(), 0,
                                 e.w.numel() // e.w.shape[0] if paired else 0])
                blocks += nb
            dev = stale[0][1].w.device
            tab = (torch.tensor(rows, dtype=torch.int64).to(dev), blocks)
            if len(_WTABLES) > 8:
                _WTABLES.clear()
            _WTABLES[tkey] = tab
        L.note_bytes((6.0 if MATH == "f16" else 8.0) * sum(e.w.numel() for _, e in stale))
        L.check(L.lib().jtsm_split_bf16_multi_f32(L.ptr(tab[0]), len(stale), C.c_long(tab[1]), int(transposed),
                                                  L.stream()), "split_bf16_multi")
        for _, e in stale:
            e.version = e.w._version


# Where a parameter's weight gradient should be WRITTEN (engine/dp.py): (data_ptr, numel) of the parameter -> a view
# into its flat gradient bucket with the parameter's strides.  A weight-gradient launch whose parameter has a slot
# (and no gradient yet this step) writes there directly, so the data-parallel exchange never copies it.
GRAD_SLOTS = {}


def grad_slot(w, params=None):
    """The bucket view for weight `w` (a parameter or a dense alias of one) if THIS launch may write it, else None:
    only the first weight-gradient launch of a parameter in a backward gets the slot — a weight used several times in
    one step (the RPN head over five pyramid levels) gets ordinary tensors for its later uses and autograd sums them
    (engine/dp.py re-arms the flags at the end of the backward).  Entries whose parameter or exchange is gone are
    dropped."""
    if not GRAD_SLOTS or w is None:
        return None
    key = (w.data_ptr(), w.numel())
    e = GRAD_SLOTS.get(key)
    if e is None:
        return None
    p = e.param()
    if p is None or e.owner() is None or p.data_ptr() != key[0]:
        GRAD_SLOTS.pop(key, None)   # stale: the model or its exchange was dropped / the parameter re-allocated
        return None
    if p.grad is not None or e.written:   # a second use, or a second backward into the same step: autograd accumulates
        return None
    e.written = True
    return e.view


# Per-shape launch facts (the ctypes shape struct, output size, FLOPs, workspace sizes, bf16x3 eligibility)
# are computed once: a training step calls the same ~90 shapes over and over.
class _Plan(object):
    __slots__ = ("s", "ref", "oh", "ow", "flops", "desc", "ws", "x3", "wgrad_big")


_PLANS = {}


def _plan(x_shape, w_shape, stride, pad, dil):
    key = (tuple(x_shape), tuple(w_shape), stride, pad, dil)
    p = _PLANS.get(key)
    if p is None:
        if len(_PLANS) >= 8192:   # (shapes with a data-dependent batch — the mask branch's foreground count — keep coming)
            _PLANS.clear()
        lib = L.lib()
        p = _Plan()
        p.wgrad_big = None
        p.s = _shape(x_shape, w_shape, stride, pad, dil)
        p.ref = C.byref(p.s)
        p.oh, p.ow = out_hw(p.s)
        p.flops = _flops(p.s)
        p.desc = _desc(p.s)
        p.x3 = tuple(bool(lib.jtsm_conv_bf16x3_eligible(p.ref, r)) and p.s.batch > 0 for r in range(3))
        p.ws = (lib.jtsm_conv_workspace_bytes(p.ref, 0), lib.jtsm_conv_workspace_bytes(p.ref, 1),
                lib.jtsm_conv_bf16x3_wgrad_workspace_bytes(p.ref), lib.jtsm_conv_bf16x3_wgrad_bias_workspace_bytes(p.ref))
        _PLANS[key] = p
    return p


# One scratch buffer per device for split-K slabs: every user runs on the caller's stream, in order.
_SCRATCH = {}


def _scratch(nbytes, device):
    if nbytes == 0:
        return None
    # (one buffer per STREAM: launches of one stream run in order, and a side stream — the queued weight gradients, the
    # semantic head — must not fold its slabs through the buffer the compute stream's contractions are using)
    key = (device, L.stream().value)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = _SCRATCH[key] = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
    return buf


_ON_SIDE_STREAM = [False]


def _x3_variant(s, role):
    """Exact bf16x3 kernel instantiation for this call (only looked up while LAUNCH_LOG is recording)."""
    if LAUNCH_LOG is None:
        return None
    v = [C.c_int() for _ in range(6)]
    L.check(L.lib().jtsm_conv_bf16x3_plan(C.byref(s), role, *[C.byref(x) for x in v]), "conv_bf16x3_plan")
    wm, wn, tm, tn, nbuf, splits = [x.value for x in v]
    np_ = ",1" if MATH == "f16" else ""    # the fp16 instantiations carry NP = 1 as their last template argument
    if role == 2:
        if nbuf == 0:
            return _Variant("igemm_x3_wgrad_halo_kernel" + ("<1>" if MATH == "f16" else ""), splits)
        return _Variant("igemm_x3_wgrad_kernel<%d,%d,%d,%d,%d%s>" % (wm, wn, tm, tn, nbuf, np_), splits)
    if nbuf == 0:   # the LDS-halo 3x3 kernel
        return _Variant("igemm_x3_halo_kernel<%s,%d,%d,%d,%d,%d%s>" % (_ROLE_NAME[role], 16 if wm == 4 else 8, wm, wn,
                                                                       tn, 21 if wm == 4 else 12, np_), splits)
    return _Variant("igemm_x3_kernel<%s,%d,%d,%d,%d,%d%s>" % (_ROLE_NAME[role], wm, wn, tm, tn, nbuf, np_), splits)


class _Variant(str):
    """Kernel instantiation name that also remembers the K slices of the call (for the finishing pass's bytes)."""
    def __new__(cls, name, splits=1):
        o = str.__new__(cls, name)
        o.splits = splits
        return o


class LaunchSpan(object):
    """hipEvents around one contraction call: before, right after its main kernel (library hook), after the call."""
    __slots__ = ("a", "mid", "b")

    def __init__(self):
        lib = L.lib()
        self.a, self.mid, self.b = lib.jtsm_event_create(), lib.jtsm_event_create(), lib.jtsm_event_create()

    def _ms(self, x, y):
        out = C.c_float()
        return out.value if L.lib().jtsm_event_elapsed_ms(C.c_void_p(x), C.c_void_p(y), C.byref(out)) == 0 else None

    def kernel_ms(self):
        """The contraction kernel alone (falls back to the whole call if the hook did not fire)."""
        ms = self._ms(self.a, self.mid)
        return ms if ms is not None else self._ms(self.a, self.b)

    def call_ms(self):
        return self._ms(self.a, self.b)

    def __del__(self):
        try:
            lib = L.lib()
            for e in (self.a, self.mid, self.b):
                lib.jtsm_event_destroy(C.c_void_p(e))
        except Exception:
            pass


def _timed(variant, flops, call, shape=None, extra_elems=0, out_elems=0, planes=False, fp32_out=True):
    """`extra_elems`: elements of the fused epilogue's extra fp32 operands (residual, accumulate, ReLU mask) — part
    of the launch's algorithmic bytes.  `out_elems` / `planes`: size of the result and whether its operand planes are
    emitted — what a split-K finishing pass moves: every slab read once, the result (+ planes, + the epilogue's extra
    operands) written / read once."""
    if LAUNCH_LOG is None:
        return call()
    splits = getattr(variant, "splits", 1)
    finish_bytes = 0.0
    if splits > 1:
        plane_b = (2.0 if MATH == "f16" else 4.0) if planes else 0.0
        finish_bytes = out_elems * (4.0 * splits + (4.0 if fp32_out else 0.0) + plane_b) + 4.0 * extra_elems
    if shape is not None and extra_elems and splits <= 1:
        shape = shape[:-1] + (shape[-1] + 4.0 * extra_elems,)
    lib = L.lib()
    span = LaunchSpan()
    st = L.stream()
    lib.jtsm_event_record(C.c_void_p(span.a), st)
    lib.jtsm_conv_set_mid_event(C.c_void_p(span.mid))
    rc = call()
    lib.jtsm_conv_set_mid_event(None)
    lib.jtsm_event_record(C.c_void_p(span.b), st)
    if variant is not None:
        variant.segment = SEGMENT
    LAUNCH_LOG.append((variant, flops, span, shape, finish_bytes))
    return rc


_ROLE_NAME = ("FWD", "DGRAD", "WGRAD")


def _numel(t):
    return 0 if t is None else t.numel()


def _variant(s, role, has_kscale=False):
    """Exact kernel instantiation the library will launch for this call (mirrors its dispatch)."""
    if LAUNCH_LOG is None:
        return None
    k, bm, bn, sp = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    L.check(L.lib().jtsm_conv_plan(C.byref(s), role, int(has_kscale), C.byref(k), C.byref(bm), C.byref(bn),
                                   C.byref(sp)), "conv_plan")
    splits = 1 if role == 2 else sp.value      # (the fp32 weight gradient accumulates with atomics: no finishing pass)
    if k.value == 0:
        return _Variant("igemm_kernel<%s,%d,%d>" % (_ROLE_NAME[role], bm.value, bn.value), splits)
    return _Variant("igemm_dma_kernel<%s,%d,%d,%d>" % (_ROLE_NAME[role], bm.value, bn.value, 2 if k.value == 1 else 1),
                    splits)


def _desc(s):
    """(shape..., algorithmic bytes): every contraction role reads two of {x, y, w} once and writes the third once,
    4 B per element either way (fp32, or a bf16 hi + lo pair)."""
    oh, ow = out_hw(s)
    elems = s.batch * (s.in_h * s.in_w * s.in_c + oh * ow * s.out_c) + s.out_c * s.in_c * s.kernel_h * s.kernel_w
    return (s.batch, s.in_h, s.in_w, s.in_c, s.out_c, s.kernel_h, s.stride, 4.0 * elems)


def _flops(s):
    oh, ow = out_hw(s)
    return 2.0 * s.batch * oh * ow * s.out_c * s.in_c * s.kernel_h * s.kernel_w


class ConvShape(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("batch", "in_h", "in_w", "in_c", "out_c", "kernel_h",
                                       "kernel_w", "stride", "pad", "dilation")]


def _shape(x_shape, w_shape, stride, pad, dil):
    n, c, h, w = x_shape
    o, i, kh, kw = w_shape
    if i != c:
        raise RuntimeError("conv2d: weight expects %d input channels, input has %d" % (i, c))
    return ConvShape(n, h, w, c, o, kh, kw, stride, pad, dil)


def out_hw(s):
    oh = (s.in_h + 2 * s.pad - s.dilation * (s.kernel_h - 1) - 1) // s.stride + 1
    ow = (s.in_w + 2 * s.pad - s.dilation * (s.kernel_w - 1) - 1) // s.stride + 1
    return oh, ow


def _cl(t):
    """channels_last storage of a logical NCHW tensor (no copy when already so)."""
    return t.contiguous(memory_format=CL)


def _check(*ts):
    L.require_gpu(*ts)
    for t in ts:
        if t is not None and t.dtype != torch.float32:
            raise RuntimeError("jtsm_amd conv kernels are float32, got %s" % t.dtype)


def conv2d_forward(x, w, stride=1, pad=0, dil=1, scale=None, bias=None, residual=None, relu=False,
                   emit_planes=False):
    _check(x, w, scale, bias, residual)
    x, w = _cl(x), _cl(w)
    pl = _plan(x.shape, w.shape, stride, pad, dil)
    s = pl.s
    y = torch.empty((s.batch, s.out_c, pl.oh, pl.ow), dtype=x.dtype, device=x.device, memory_format=CL)
    if residual is not None:
        residual = _cl(residual)
        assert residual.shape == y.shape
    nbytes = pl.ws[0]
    ws = _scratch(nbytes, x.device)
    lib = L.lib()
    if MATH != "f32" and pl.x3[0]:
        xh, xl = _hl(planes_of(x))
        wbuf = _weight_planes(w)
        wh, wl = _hl(wbuf)
        ybuf = _planes_buf(y.numel(), y.device) if (emit_planes and s.out_c % 8 == 0) else None
        yh, yl = _hl(ybuf)
        if MATH == "f16":
            L.check(_timed(_x3_variant(s, 0), pl.flops, lambda: lib.jtsm_conv2d_forward_f16(
                xh, wh, L.ptr(y), yh, pl.ref, L.ptr(scale), L.ptr(bias), L.ptr(residual), int(bool(relu)),
                L.ptr(ws), C.c_size_t(nbytes), L.stream()), pl.desc, _numel(residual), y.numel(), ybuf is not None),
                    "conv2d_forward_f16")
        else:
            L.check(_timed(_x3_variant(s, 0), pl.flops, lambda: lib.jtsm_conv2d_forward_bf16x3(
                xh, xl, wh, wl, L.ptr(y), yh, yl, pl.ref, L.ptr(scale), L.ptr(bias), L.ptr(residual), int(bool(relu)),
                L.ptr(ws), C.c_size_t(nbytes), L.stream()), pl.desc, _numel(residual), y.numel(), ybuf is not None),
                    "conv2d_forward_bf16x3")
        if ybuf is not None:
            planes_put(y, ybuf)
        return y
    L.check(_timed(_variant(s, 0), pl.flops, lambda: lib.jtsm_conv2d_forward_f32(
        L.ptr(x), L.ptr(w), L.ptr(y), pl.ref, L.ptr(scale), L.ptr(bias), L.ptr(residual), int(bool(relu)),
        L.ptr(ws), C.c_size_t(nbytes), L.stream()), pl.desc, _numel(residual), y.numel()), "conv2d_forward")
    return y


def conv2d_backward_data(dy, w, x_shape, stride=1, pad=0, dil=1, kscale=None, accumulate=None,
                         relu_mask=None, emit_planes=False, into=None):
    """`into`: a gradient map of x's shape (channels-last fp32) that this data gradient is ADDED to in place — the
    epilogue's residual operand and its result are the same piece of the same thread (layers/grad_fan.py)."""
    _check(dy, w, kscale, accumulate, relu_mask)
    dy, w = _cl(dy), _cl(w)
    pl = _plan(x_shape, w.shape, stride, pad, dil)
    s = pl.s
    if into is not None:
        assert accumulate is None and tuple(into.shape) == tuple(x_shape) and into.is_contiguous(memory_format=CL)
        planes_forget(into)
        dx = accumulate = into
    else:
        dx = torch.empty(tuple(x_shape), dtype=dy.dtype, device=dy.device, memory_format=CL)
    if accumulate is not None:
        accumulate = _cl(accumulate)
    if relu_mask is not None:
        relu_mask = _cl(relu_mask)
    nbytes = pl.ws[1]
    ws = _scratch(nbytes, dy.device)
    lib = L.lib()
    if MATH != "f32" and pl.x3[1]:
        gh, gl = _hl(planes_of(dy, grad=True))
        wbuf = _weight_planes(w, True, kscale)   # the per-row scale rides along in the transposing split
        wh, wl = _hl(wbuf)
        scatter = s.kernel_h == 1 and s.kernel_w == 1 and s.pad == 0 and s.stride > 1 and \
            (accumulate is None or into is not None) and relu_mask is None
        dbuf = _planes_buf(dx.numel(), dx.device) if (emit_planes and s.in_c % 8 == 0 and not scatter) else None
        dh, dl = _hl(dbuf)
        if MATH == "f16":
            L.check(_timed(_x3_variant(s, 1), pl.flops, lambda: lib.jtsm_conv2d_backward_data_f16(
                gh, wh, L.ptr(dx), dh, pl.ref, L.ptr(accumulate), L.ptr(relu_mask), GRAD_SHIFT, L.ptr(ws),
                C.c_size_t(nbytes), L.stream()), pl.desc, _numel(accumulate) + _numel(relu_mask), dx.numel(),
                    dbuf is not None), "conv2d_backward_data_f16")
        else:
            L.check(_timed(_x3_variant(s, 1), pl.flops, lambda: lib.jtsm_conv2d_backward_data_bf16x3(
                gh, gl, wh, wl, L.ptr(dx), dh, dl, pl.ref, L.ptr(accumulate), L.ptr(relu_mask), L.ptr(ws),
                C.c_size_t(nbytes), L.stream()), pl.desc, _numel(accumulate) + _numel(relu_mask), dx.numel(),
                    dbuf is not None), "conv2d_backward_data_bf16x3")
        if dbuf is not None:
            planes_put(dx, dbuf)
        return dx
    L.check(_timed(_variant(s, 1, kscale is not None), pl.flops, lambda: lib.jtsm_conv2d_backward_data_f32(
        L.ptr(dy), L.ptr(w), L.ptr(dx), pl.ref, L.ptr(kscale), L.ptr(accumulate), L.ptr(relu_mask),
        L.ptr(ws), C.c_size_t(nbytes), L.stream()), pl.desc, _numel(accumulate) + _numel(relu_mask), dx.numel()),
            "conv2d_backward_data")
    return dx


def conv2d_backward_weight(dy, x, w_shape, stride=1, pad=0, dil=1, row_scale=None, out=None, w=None, bias_out=None):
    """`out`: accumulate into this tensor.  `w`: the weight being differentiated — when the data-parallel exchange
    has registered a gradient slot for it, the result is written there (a fresh result, not an accumulation).
    `bias_out` (out_c,): also the bias gradient sum_pixels dy — inside the same contraction in the plane arithmetics
    (no second pass over dy), by a channel_sum pass otherwise."""
    _check(dy, x, row_scale)
    dy, x = _cl(dy), _cl(x)
    pl = _plan(x.shape, w_shape, stride, pad, dil)
    s = pl.s
    lib = L.lib()
    slot = grad_slot(w) if out is None else None
    if slot is not None:   # same bytes as the OHWI result the kernels write (the parameter is dense, channels_last)
        n, taps = w_shape[0], w_shape[2] * w_shape[3]
        slot = slot.as_strided(tuple(w_shape), (taps * w_shape[1], 1, w_shape[3] * w_shape[1], w_shape[1]))
    if MATH != "f32" and pl.x3[2]:
        gh, gl = _hl(planes_of(dy, grad=True))
        xh, xl = _hl(planes_of(x))
        fresh = out is None
        if fresh:   # deterministic slab kernel: writes every element, nothing to clear
            out = slot if slot is not None else torch.empty(tuple(w_shape), dtype=x.dtype, device=x.device,
                                                            memory_format=CL)
        if bias_out is not None and BIAS_IN_WGRAD and _wgrad_bias_fits(pl):
            _wgrad_bias_call(pl, gh, gl, xh, xl, out, bias_out, row_scale, fresh, x.device)
            return out
        if bias_out is not None:
            from .elementwise import channel_sum
            bias_out.copy_(channel_sum(dy))
        nbytes = pl.ws[2]
        ws = _scratch(nbytes, x.device)
        if MATH == "f16":
            L.check(_timed(_x3_variant(s, 2), pl.flops, lambda: lib.jtsm_conv2d_backward_weight_f16(
                gh, xh, L.ptr(out), pl.ref, L.ptr(row_scale), int(fresh), GRAD_SHIFT, L.ptr(ws), C.c_size_t(nbytes),
                L.stream()), pl.desc, 0, out.numel()), "conv2d_backward_weight_f16")
        else:
            L.check(_timed(_x3_variant(s, 2), pl.flops, lambda: lib.jtsm_conv2d_backward_weight_bf16x3(
                gh, gl, xh, xl, L.ptr(out), pl.ref, L.ptr(row_scale), int(fresh), L.ptr(ws), C.c_size_t(nbytes),
                L.stream()), pl.desc, 0, out.numel()), "conv2d_backward_weight_bf16x3")
        return out
    if bias_out is not None:
        from .elementwise import channel_sum
        bias_out.copy_(channel_sum(dy))
    zero = False   # cleared here (not inside the timed launch) so per-launch timings are kernel-only
    if out is None:
        out = slot.zero_() if slot is not None else \
            torch.empty(tuple(w_shape), dtype=x.dtype, device=x.device, memory_format=CL).zero_()
    L.check(_timed(_variant(s, 2), pl.flops, lambda: lib.jtsm_conv2d_backward_weight_f32(
        L.ptr(dy), L.ptr(x), L.ptr(out), pl.ref, L.ptr(row_scale), int(zero), L.stream()), pl.desc),
            "conv2d_backward_weight")
    return out


# ---- ConvTranspose2d(kernel 2, stride 2, padding 0): the mask heads' upsampler -----------------------------------
def conv_transpose2x2_ok(x, w):
    """The native path: planes arithmetic, channels_last (in, out, 2, 2) weight, 32-channel multiples."""
    return (MATH != "f32" and w.dim() == 4 and tuple(w.shape[2:]) == (2, 2) and w.shape[0] % 32 == 0 and
            w.shape[1] % 32 == 0 and w.permute(0, 2, 3, 1).is_contiguous() and x.shape[0] > 0)


def _ct_desc(n, h, w, i, o):
    elems = n * h * w * (i + 4 * o) + 4 * i * o
    return (n, h, w, i, 4 * o, 1, 1, 4.0 * elems)


def conv_transpose2x2_forward(x, w, bias=None, relu=False, emit_planes=False):
    """y (N, out, 2H, 2W) channels_last = relu?(conv_transpose2d(x, w, stride 2) + bias): one GEMM of (N*H*W) x
    (4*out) whose epilogue writes each row's four column groups to the four output pixels."""
    _check(x, w, bias)
    x = _cl(x)
    n, i, h, wd = x.shape
    o = w.shape[1]
    y = torch.empty((n, o, 2 * h, 2 * wd), dtype=x.dtype, device=x.device, memory_format=CL)
    xh, xl = _hl(planes_of(x))
    w1 = w.as_strided((i, 4 * o, 1, 1), (4 * o, 1, 1, 1))       # the same memory as a 1x1 weight: [in][(dy,dx,out)]
    wth, wtl = _hl(_weight_planes(w1, True))
    ybuf = _planes_buf(y.numel(), y.device) if emit_planes else None
    yh, yl = _hl(ybuf)
    lib = L.lib()
    flops = 2.0 * n * h * wd * i * 4 * o
    var = _x3_variant(_plan((n, i, h, wd), (4 * o, i, 1, 1), 1, 0, 1).s, 0)
    var = _Variant(str(var), 1) if var is not None else None      # (the pixel-shuffle epilogue never splits K)
    if MATH == "f16":
        L.check(_timed(var, flops, lambda: lib.jtsm_conv_transpose2x2_forward_f16(
            xh, wth, L.ptr(y), yh, n, h, wd, i, o, L.ptr(bias), int(bool(relu)), L.stream()),
            _ct_desc(n, h, wd, i, o), 0, y.numel(), ybuf is not None), "conv_transpose2x2_forward_f16")
    else:
        L.check(_timed(var, flops, lambda: lib.jtsm_conv_transpose2x2_forward_bf16x3(
            xh, xl, wth, wtl, L.ptr(y), yh, yl, n, h, wd, i, o, L.ptr(bias), int(bool(relu)), L.stream()),
            _ct_desc(n, h, wd, i, o), 0, y.numel(), ybuf is not None), "conv_transpose2x2_forward_bf16x3")
    if ybuf is not None:
        planes_put(y, ybuf)
    return y


def conv_transpose2x2_backward_data(g, w, relu_mask=None, emit_planes=False):
    """dx (N, in, H, W) of conv_transpose2x2_forward, kept where relu_mask > 0: the forward role of the 2x2 / stride-2
    convolution whose OHWI weight is the parameter's memory."""
    _check(g, w, relu_mask)
    g = _cl(g)
    n, o, h2, w2 = g.shape
    i = w.shape[0]
    h, wd = h2 // 2, w2 // 2
    pl = _plan(g.shape, (i, o, 2, 2), 2, 0, 1)
    dx = torch.empty((n, i, h, wd), dtype=g.dtype, device=g.device, memory_format=CL)
    if relu_mask is not None:
        relu_mask = _cl(relu_mask)
        assert relu_mask.shape == dx.shape
    gh, gl = _hl(planes_of(g, grad=True))
    wh, wl = _hl(_weight_planes(w))
    dbuf = _planes_buf(dx.numel(), dx.device) if emit_planes else None
    dh, dl = _hl(dbuf)
    nbytes = pl.ws[0]
    ws = _scratch(nbytes, g.device)
    lib = L.lib()
    if MATH == "f16":
        L.check(_timed(_x3_variant(pl.s, 0), pl.flops, lambda: lib.jtsm_conv_transpose2x2_backward_data_f16(
            gh, wh, L.ptr(dx), dh, n, h, wd, i, o, L.ptr(relu_mask), None, GRAD_SHIFT, L.ptr(ws), C.c_size_t(nbytes),
            L.stream()), pl.desc, _numel(relu_mask), dx.numel(), dbuf is not None), "conv_transpose2x2_backward_data_f16")
    else:
        L.check(_timed(_x3_variant(pl.s, 0), pl.flops, lambda: lib.jtsm_conv_transpose2x2_backward_data_bf16x3(
            gh, gl, wh, wl, L.ptr(dx), dh, dl, n, h, wd, i, o, L.ptr(relu_mask), None, L.ptr(ws), C.c_size_t(nbytes),
            L.stream()), pl.desc, _numel(relu_mask), dx.numel(), dbuf is not None),
                "conv_transpose2x2_backward_data_bf16x3")
    if dbuf is not None:
        planes_put(dx, dbuf)
    return dx


def conv_transpose2x2_backward_weight(g, x, w):
    """dW in the parameter's own (channels_last) memory: the weight gradient of that 2x2 / stride-2 convolution with
    the roles swapped — its 'dy' is x, its input is g."""
    i, o = w.shape[0], w.shape[1]
    if MATH == "f16":   # exactly one operand may carry 2^GRAD_SHIFT: g's planes do, so x's must be plain
        planes_of(x)
        planes_of(g, grad=True)
    return conv2d_backward_weight(x, g, (i, o, 2, 2), 2, 0, 1, w=w)


# ---- the same contractions on PLANE BUFFERS: a chain (layers/fused_blocks.py: the mask tower) that keeps its
# activations and gradients as operand planes only — no fp32 copy is written, ReLU gates are read from the hi plane
class PlaneTensor(object):
    """Planes buffer (layers/conv.py: _planes_buf) of a logical (N, C, H, W) channels_last tensor."""
    __slots__ = ("buf", "shape")

    def __init__(self, buf, shape):
        self.buf, self.shape = buf, tuple(shape)

    @property
    def numel(self):
        n = 1
        for d in self.shape:
            n *= d
        return n

    @property
    def device(self):
        return self.buf.device

    @staticmethod
    def of(t, grad=False):
        return PlaneTensor(planes_of(_cl(t), grad), t.shape)

    @staticmethod
    def empty(shape, device):
        n = 1
        for d in shape:
            n *= d
        return PlaneTensor(_planes_buf(n, device), shape)


def planes_forward(x, w, stride=1, pad=0, dil=1, bias=None, relu=False, fp32=False, scale=None, residual=None,
                   residual_plane=None):
    """x: PlaneTensor -> PlaneTensor of relu?(conv(x, w) + bias) (fp32=False: planes only, no fp32 copy is written), or
    the fp32 result alone (fp32=True: a chain's last layer, read by a loss).  residual_plane (fp16 arithmetic only): the
    residual as a PlaneTensor — a block input that exists as its operand plane alone."""
    pl = _plan(x.shape, w.shape, stride, pad, dil)
    s = pl.s
    if not pl.x3[0]:
        raise RuntimeError("planes_forward: shape is not eligible for the plane arithmetic")
    oshape = (s.batch, s.out_c, pl.oh, pl.ow)
    both = fp32 == "both"            # -> (y fp32, PlaneTensor)
    y = torch.empty(oshape, dtype=torch.float32, device=x.device, memory_format=CL) if fp32 else None
    yp = PlaneTensor.empty(oshape, x.device) if (both or not fp32) else None
    xh, xl = _hl(x.buf)
    wh, wl = _hl(_weight_planes(_cl(w)))
    yh, yl = _hl(yp.buf if yp is not None else None)
    nbytes = pl.ws[0]
    ws = _scratch(nbytes, x.device)
    lib = L.lib()
    n_out = s.batch * s.out_c * pl.oh * pl.ow
    if residual is not None:
        residual = _cl(residual)
    if residual_plane is not None:
        assert MATH == "f16" and residual is None and tuple(residual_plane.shape) == oshape
        rh = _hl(residual_plane.buf)[0]
        L.check(_timed(_x3_variant(s, 0), pl.flops, lambda: lib.jtsm_conv2d_forward_res16_f16(
            xh, wh, L.ptr(y), yh, pl.ref, L.ptr(scale), L.ptr(bias), rh, int(bool(relu)), L.ptr(ws),
            C.c_size_t(nbytes), L.stream()), pl.desc, 0.5 * n_out, n_out, yp is not None, bool(fp32)),
                "conv2d_forward_res16_f16")
    elif MATH == "f16":
        L.check(_timed(_x3_variant(s, 0), pl.flops, lambda: lib.jtsm_conv2d_forward_f16(
            xh, wh, L.ptr(y), yh, pl.ref, L.ptr(scale), L.ptr(bias), L.ptr(residual), int(bool(relu)), L.ptr(ws),
            C.c_size_t(nbytes), L.stream()), pl.desc, _numel(residual), n_out, yp is not None, bool(fp32)),
                "conv2d_forward_f16")
    else:
        L.check(_timed(_x3_variant(s, 0), pl.flops, lambda: lib.jtsm_conv2d_forward_bf16x3(
            xh, xl, wh, wl, L.ptr(y), yh, yl, pl.ref, L.ptr(scale), L.ptr(bias), L.ptr(residual), int(bool(relu)),
            L.ptr(ws), C.c_size_t(nbytes), L.stream()), pl.desc, _numel(residual), n_out, yp is not None, bool(fp32)),
                "conv2d_forward_bf16x3")
    if both:
        return y, yp
    return y if fp32 else yp


def _colsum_partials(pl, role, width, device):
    """(rows, width) buffer for the per-row-tile column sums of a data-gradient launch, or None when the shape has none."""
    rows = L.lib().jtsm_conv_bf16x3_colsum_rows(C.byref(pl.s), role)
    return torch.empty((rows, width), dtype=torch.float32, device=device) if rows > 0 else None


class ColsumBatch(object):
    """Bias gradients taken in data-gradient epilogues (planes_backward_data(bias_out=batch.slot(key))): the launches
    leave per-row-tile partial sums, `finish()` adds them up — all layers of one width in ONE launch — and returns
    {key: (width,) tensor}."""

    def __init__(self):
        self.items = []          # (key, partial (rows, width))

    def slot(self, key):
        return _ColsumSlot(self, key)

    def finish(self):
        out = {}
        by_width = {}
        for key, part in self.items:
            by_width.setdefault(part.shape[1], []).append((key, part))
        for width, group in by_width.items():
            for i0 in range(0, len(group), 16):
                chunk = group[i0:i0 + 16]
                sums = torch.empty((len(chunk), width), dtype=torch.float32, device=chunk[0][1].device)
                ptrs = (C.c_void_p * len(chunk))(*[p.data_ptr() for _, p in chunk])
                rows = (C.c_int * len(chunk))(*[int(p.shape[0]) for _, p in chunk])
                L.check(L.lib().jtsm_colsum_fold_f32(ptrs, rows, len(chunk), width, L.ptr(sums), L.stream()), "colsum_fold")
                for k, (key, _) in enumerate(chunk):
                    out[key] = sums[k]
        self.items = []
        return out


class _ColsumSlot(object):
    __slots__ = ("batch", "key")

    def __init__(self, batch, key):
        self.batch, self.key = batch, key

    def take(self, part):
        self.batch.items.append((self.key, part))


def planes_backward_data(g, w, x_shape, stride=1, pad=0, dil=1, gate=None, fp32=False, accumulate=None,
                         row_scale=None, kscale=None, both=False, into=None, bias_out=None, accumulate_plane=None):
    """bias_out (a ColsumBatch slot): ALSO the column sums of the finished (gated) result — the bias gradient of the
    layer whose output gradient this call produces — taken in the epilogue that writes it.  Returns (result, True)
    then, or (result, False) when this shape cannot (the caller sums the planes afterwards)."""
    if bias_out is not None:
        assert not fp32 and not both and accumulate is None and into is None
        return _planes_backward_data_colsum(g, w, x_shape, stride, pad, dil, gate, kscale, bias_out, row_scale)
    return _planes_backward_data(g, w, x_shape, stride, pad, dil, gate, fp32, accumulate, row_scale, kscale, both, into,
                                 accumulate_plane)


def _planes_backward_data_colsum(g, w, x_shape, stride, pad, dil, gate, kscale, bias_out, row_scale=None):
    pl = _plan(x_shape, w.shape, stride, pad, dil)
    part = _colsum_partials(pl, 1, x_shape[1], g.device) if pl.x3[1] else None
    if part is None:
        return _planes_backward_data(g, w, x_shape, stride, pad, dil, gate, False, None, row_scale, kscale, False,
                                     None), False
    dp = PlaneTensor.empty(x_shape, g.device)
    gh, gl = _hl(g.buf)
    wh, wl = _hl(_weight_planes(_cl(w), True, kscale))
    dh, dl = _hl(dp.buf)
    gate_h = _hl(gate.buf)[0] if gate is not None else None
    nbytes = pl.ws[1]
    ws = _scratch(nbytes, g.device)
    lib = L.lib()
    extra = 0.5 * dp.numel if gate is not None else 0
    if MATH == "f16":
        L.check(_timed(_x3_variant(pl.s, 1), pl.flops, lambda: lib.jtsm_conv2d_backward_data_colsum_f16(
            gh, wh, None, dh, pl.ref, L.ptr(row_scale), None, None, gate_h, GRAD_SHIFT, L.ptr(part), L.ptr(ws),
            C.c_size_t(nbytes), L.stream()), pl.desc, extra, dp.numel, True, False), "conv2d_backward_data_colsum_f16")
    else:
        L.check(_timed(_x3_variant(pl.s, 1), pl.flops, lambda: lib.jtsm_conv2d_backward_data_colsum_bf16x3(
            gh, gl, wh, wl, None, dh, dl, pl.ref, L.ptr(row_scale), None, None, gate_h, L.ptr(part), L.ptr(ws),
            C.c_size_t(nbytes), L.stream()), pl.desc, extra, dp.numel, True, False), "conv2d_backward_data_colsum_bf16x3")
    bias_out.take(part)
    return dp, True


def _planes_backward_data(g, w, x_shape, stride=1, pad=0, dil=1, gate=None, fp32=False, accumulate=None,
                          row_scale=None, kscale=None, both=False, into=None, accumulate_plane=None):
    """... kscale: per-output-channel factor folded into the weight rows (FrozenBN).  both: fp32 AND planes."""
    """g: PlaneTensor of the output gradient (gradient planes: times 2^GRAD_SHIFT in fp16 mode); gate: PlaneTensor of
    the ReLU output the result is gated by (or None) -> the input gradient as fp32 tensor (fp32=True) or PlaneTensor."""
    pl = _plan(x_shape, w.shape, stride, pad, dil)
    s = pl.s
    if not pl.x3[1]:
        raise RuntimeError("planes_backward_data: shape is not eligible for the plane arithmetic")
    fp32 = fp32 or both
    if into is not None:   # add into an existing gradient map in place (see conv2d_backward_data)
        assert fp32 and not both and accumulate is None and gate is None and tuple(into.shape) == tuple(x_shape) and \
            into.is_contiguous(memory_format=CL)
        planes_forget(into)
        dx = accumulate = into
    else:
        dx = torch.empty(tuple(x_shape), dtype=torch.float32, device=g.device, memory_format=CL) if fp32 else None
    dp = PlaneTensor.empty(x_shape, g.device) if (both or not fp32) else None
    gh, gl = _hl(g.buf)
    wh, wl = _hl(_weight_planes(_cl(w), True, kscale))
    dh, dl = _hl(dp.buf if dp is not None else None)
    gate_h = _hl(gate.buf)[0] if gate is not None else None
    nbytes = pl.ws[1]
    ws = _scratch(nbytes, g.device)
    lib = L.lib()
    n_in = 1
    for d in x_shape:
        n_in *= d
    extra = (0.5 * n_in if gate is not None else 0) + _numel(accumulate)     # (the gate plane: 2 bytes per element)
    if accumulate is not None:
        accumulate = _cl(accumulate)
    if accumulate_plane is not None:
        # another gradient term as an fp16 plane (times 2^GRAD_SHIFT): the shortcut path of a chain whose gradient
        # stream has no fp32 copy
        assert MATH == "f16" and accumulate is None and into is None and tuple(accumulate_plane.shape) == tuple(x_shape)
        ah = _hl(accumulate_plane.buf)[0]
        L.check(_timed(_x3_variant(s, 1), pl.flops, lambda: lib.jtsm_conv2d_backward_data_acc16_f16(
            gh, wh, L.ptr(dx), dh, pl.ref, L.ptr(row_scale), ah, gate_h, GRAD_SHIFT, L.ptr(ws), C.c_size_t(nbytes),
            L.stream()), pl.desc, extra + 0.5 * n_in, n_in, dp is not None, fp32), "conv2d_backward_data_acc16_f16")
    elif MATH == "f16":
        L.check(_timed(_x3_variant(s, 1), pl.flops, lambda: lib.jtsm_conv2d_backward_data_ex_f16(
            gh, wh, L.ptr(dx), dh, pl.ref, L.ptr(row_scale), L.ptr(accumulate), None, gate_h, GRAD_SHIFT, L.ptr(ws), C.c_size_t(nbytes), L.stream()),
            pl.desc, extra, n_in, dp is not None, fp32), "conv2d_backward_data_ex_f16")
    else:
        L.check(_timed(_x3_variant(s, 1), pl.flops, lambda: lib.jtsm_conv2d_backward_data_ex_bf16x3(
            gh, gl, wh, wl, L.ptr(dx), dh, dl, pl.ref, L.ptr(row_scale), L.ptr(accumulate), None, gate_h, L.ptr(ws), C.c_size_t(nbytes),
            L.stream()), pl.desc, extra, n_in, dp is not None, fp32), "conv2d_backward_data_ex_bf16x3")
    if both:
        return dx, dp
    return dx if fp32 else dp


def planes_backward_weight(g, x, w, stride=1, pad=0, dil=1, w_shape=None, row_scale=None, bias_out=None):
    """dW from the planes of the output gradient g and of the input x; written into the parameter's gradient slot when
    the data-parallel exchange registered one."""
    w_shape = tuple(w.shape) if w_shape is None else tuple(w_shape)
    pl = _plan(x.shape, w_shape, stride, pad, dil)
    s = pl.s
    if not pl.x3[2]:
        raise RuntimeError("planes_backward_weight: shape is not eligible for the plane arithmetic")
    slot = grad_slot(w)
    if slot is not None:
        taps = w_shape[2] * w_shape[3]
        slot = slot.as_strided(w_shape, (taps * w_shape[1], 1, w_shape[3] * w_shape[1], w_shape[1]))
    out = slot if slot is not None else torch.empty(w_shape, dtype=torch.float32, device=x.device, memory_format=CL)
    gh, gl = _hl(g.buf)
    xh, xl = _hl(x.buf)
    if bias_out is not None and BIAS_IN_WGRAD and _wgrad_bias_fits(pl):   # db beside dW, from the same gradient planes
        _wgrad_bias_call(pl, gh, gl, xh, xl, out, bias_out, row_scale, True, x.device)
        return out
    if bias_out is not None:
        bias_out.copy_(planes_channel_sum(g))
    nbytes = pl.ws[2]
    ws = _scratch(nbytes, x.device)
    lib = L.lib()
    if MATH == "f16":
        L.check(_timed(_x3_variant(s, 2), pl.flops, lambda: lib.jtsm_conv2d_backward_weight_f16(
            gh, xh, L.ptr(out), pl.ref, L.ptr(row_scale), 1, GRAD_SHIFT, L.ptr(ws), C.c_size_t(nbytes), L.stream()),
            pl.desc, 0, out.numel()), "conv2d_backward_weight_f16")
    else:
        L.check(_timed(_x3_variant(s, 2), pl.flops, lambda: lib.jtsm_conv2d_backward_weight_bf16x3(
            gh, gl, xh, xl, L.ptr(out), pl.ref, L.ptr(row_scale), 1, L.ptr(ws), C.c_size_t(nbytes), L.stream()),
            pl.desc, 0, out.numel()), "conv2d_backward_weight_bf16x3")
    return out


# ---- deferred, grouped weight gradients --------------------------------------------------------------------------
# A weight gradient is not on the backward's critical path (only the optimizer and the gradient exchange read it), and
# res3 / res4 / res5 repeat three layer shapes 3-6 times.  Launched where autograd reaches them, each of those small
# contractions is cut into ~32 pixel slices to fill the chip and its slabs are written and folded again.  With
# DEFER_WGRAD the fused bottleneck node only QUEUES its weight gradients (operand planes kept alive) and hands autograd
# nothing; flush_deferred_weight_gradients() — called by the first block of every stage (the last of the stage to run
# in a backward), by the gradient exchange before it closes its buckets, and at the end of the backward — launches the
# queued layers of one shape as ONE group (jtsm_conv2d_backward_weight_group_*: a sixth of the slices, one finishing
# launch), stores the results as the parameters' .grad (accumulating into an existing one) and runs their
# post-accumulate hooks, which is all AccumulateGrad would have done.
# Only for leaf parameters reached through `.backward()`: `torch.autograd.grad(..., inputs=[weight])` must switch it off
# (defer_weight_gradients(False)), since the node returns None for a queued weight.
DEFER_WGRAD = os.environ.get("JTSM_DEFER_WGRAD", "1") != "0"
GROUP_MAX = 8
_DEFERRED = []
_DEFERRED_SEGMENT = {}     # id(weight) -> SEGMENT at queueing time (measurement only)
_DEFERRED_PENDING = {}     # id(weight) -> queued launches not delivered yet (engine/dp.py ignores autograd's hook for them)
STALE_DROPPED = [0]        # (tests) queued launches of an aborted backward dropped at the next forward


def deferred_pending(w):
    """True while a weight gradient of `w` sits in the queue: autograd still runs the parameter's post-accumulate hooks
    when the node handed it nothing (torch 2.10), and whatever `.grad` holds then is NOT this backward's gradient."""
    return _DEFERRED_PENDING.get(id(w), 0) > 0


def _drop_stale_deferred():
    """Start of a forward (planes_clear): a backward that aborted (an exception, out of memory) never ran its engine
    callbacks — its queue would pin that step's operand planes and be added into the NEXT step's gradients."""
    global _DEFERRED
    if _DEFERRED:
        STALE_DROPPED[0] += len(_DEFERRED)
        _DEFERRED = []
    _DEFERRED_PENDING.clear()


CLEAR_HOOKS.append(_drop_stale_deferred)


def defer_weight_gradients(on):
    global DEFER_WGRAD
    flush_deferred_weight_gradients()
    DEFER_WGRAD = bool(on)


def planes_backward_weight_deferred(g, x, w, stride=1, pad=0, dil=1, row_scale=None):
    """Queue dW of a leaf parameter for the grouped launch and return None — or, when deferral does not apply (switched
    off, not a leaf parameter, no backward in progress), compute it now and return it."""
    if not (DEFER_WGRAD and MATH != "f32" and w.is_leaf and w.requires_grad and w._base is None):
        return planes_backward_weight(g, x, w, stride, pad, dil, row_scale=row_scale)
    if not _DEFERRED:
        # the queue is empty: (re-)arm the end-of-backward flush for THIS backward pass.  Keyed to the queue, not to a
        # sticky flag — a flag left set by an aborted backward would never queue a callback again; a second callback in
        # one backward (after a stage's own flush emptied the queue) finds nothing to do
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_final_flush)
        except RuntimeError:     # not inside a backward pass: nothing would flush the queue
            return planes_backward_weight(g, x, w, stride, pad, dil, row_scale=row_scale)
    _DEFERRED.append((g, x, w, stride, pad, dil, row_scale))
    _DEFERRED_SEGMENT[id(w)] = SEGMENT
    _DEFERRED_PENDING[id(w)] = _DEFERRED_PENDING.get(id(w), 0) + 1
    return None


def _deliver_grad(w, dw):
    """What AccumulateGrad does for a leaf: store (or add) the gradient, then the post-accumulate hooks."""
    if dw.dim() == 4 and dw.stride() != w.stride() and dw.shape[2] == 1 and dw.shape[3] == 1:
        dw = dw.as_strided(w.shape, w.stride())      # (a 1x1 weight: both orders are the same bytes)
    if w.grad is None:
        w.grad = dw
    else:
        w.grad.add_(dw)
    left = _DEFERRED_PENDING.get(id(w), 1) - 1
    if left > 0:                 # a weight queued more than once in this backward: the hooks run behind the last term
        _DEFERRED_PENDING[id(w)] = left
        return
    _DEFERRED_PENDING.pop(id(w), None)
    hooks = getattr(w, "_post_accumulate_grad_hooks", None)
    if hooks:
        for hook in list(hooks.values()):
            hook(w)


# The queued weight gradients on a SIDE stream (JTSM_WGRAD_STREAM=0: off): nothing in the backward waits for them, so a
# stage's grouped launches can run beside the next stage's (short, latency-bound) data gradients instead of in front of
# them.  Under post-accumulate hooks only when they are the gradient exchange's alone (engine/dp.py: its collectives wait
# for every producer stream; `_hooks_allow_side_stream`) and the weight is used once in the forward; the end of the
# backward pass makes the compute stream wait for the side stream.  Same-process A/B
# (tools/sweeps/wgrad_stream_ab.py, 4 x 25 steps each way): 21.74-22.10 -> 21.54-21.64 ms per step.  (Round 1 measured a
# weight gradient beside ITS OWN layer's data gradient as no gain — two large kernels competing for operand delivery;
# here a stage's grouped gradients run beside the next stage's short layers, which leave the matrix pipes idle.)
WGRAD_STREAM = os.environ.get("JTSM_WGRAD_STREAM", "1") != "0"
_WGRAD_SIDE = {}
# Every stream besides the compute stream on which this package produces gradients (this file's weight-gradient stream,
# meta_arch/mcnn.py's semantic-head stream): the gradient exchange (engine/dp.py) makes its collectives wait for all of
# them, which is what lets the side streams stay on under it.
PRODUCER_STREAMS = []


def register_producer_stream(stream):
    if all(s is not stream for s in PRODUCER_STREAMS):
        PRODUCER_STREAMS.append(stream)


def _wgrad_side_stream(device):
    st = _WGRAD_SIDE.get(device)
    if st is None:
        st = _WGRAD_SIDE[device] = torch.cuda.Stream(device=device)
        register_producer_stream(st)
    return st


# Uses of a weight by _ConvFused nodes in the current forward (cleared with the plane cache): a weight used more than once
# (an RPN head over five pyramid levels) keeps autograd's accumulation under hooks — AccumulateGrad runs a parameter's
# hooks once, behind the SUM of its terms; a per-use delivery would run them behind the first.
_FORWARD_USES = {}
CLEAR_HOOKS.append(_FORWARD_USES.clear)


# Callables run ONCE per backward pass, when the first contraction node of the FPN starts its backward: autograd runs the
# ready node with the highest sequence number, and every node of the heads (box, mask, semantic) was created behind the
# pyramid — so all of them have run, i.e. every head gradient is delivered or queued.  solver/build.py hangs the heads'
# share of the optimizer step here (on the weight-gradient side stream, beside the FPN's and the backbone's backward).
HEADS_DONE_HOOKS = []
_HEADS_DONE = [False]
CLEAR_HOOKS.append(lambda: _HEADS_DONE.__setitem__(0, False))


def _heads_done():
    if _HEADS_DONE[0] or not HEADS_DONE_HOOKS:
        return
    _HEADS_DONE[0] = True
    for hook in list(HEADS_DONE_HOOKS):
        hook()


def queue_side_stream_join():
    """Make sure the end of this backward pass joins the weight-gradient side stream into the compute stream (callers that
    put work on that stream themselves).  False outside a backward pass."""
    if not _JOIN_QUEUED[0]:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_final_flush)
        except RuntimeError:
            return False
        _JOIN_QUEUED[0] = True
    return True


def _hooks_allow_side_stream(p):
    """A parameter's post-accumulate hooks do not tie its gradient to the compute stream: it has none, or only the
    gradient exchange's (engine/dp.py marks its parameters: its collectives wait for every producer stream)."""
    hooks = getattr(p, "_post_accumulate_grad_hooks", None)
    return not hooks or (len(hooks) == 1 and getattr(p, "_jtsm_exchange_waits_for_producers", False))


def join_wgrad_stream():
    """The compute stream waits for the weight gradients launched on the side stream (end of the backward pass)."""
    for device, st in _WGRAD_SIDE.items():
        torch.cuda.current_stream(device).wait_stream(st)


def _final_flush():
    _JOIN_QUEUED[0] = False
    flush_deferred_weight_gradients()
    join_wgrad_stream()


_JOIN_QUEUED = [False]     # an end-of-backward join is queued for this backward pass (reset with the plane cache too)
CLEAR_HOOKS.append(lambda: _JOIN_QUEUED.__setitem__(0, False))


def side_weight_gradients(params, compute, operands=()):
    """Weight (and bias) gradients that nothing in the backward waits for, computed NOW but on the weight-gradient side
    stream: `compute()` launches the contraction(s) and returns one gradient per entry of `params` (leaf parameters; a
    None entry is skipped); the gradients are delivered here as AccumulateGrad would (the caller returns None to autograd
    for them).  `operands`: buffers the launches read that may be freed before the side stream is done.  False: not
    applicable (switched off, a parameter that is a view / carries hooks / needs no gradient, no backward in progress) —
    the caller computes the gradients the ordinary way."""
    live = [p for p in params if p is not None]
    if not (WGRAD_STREAM and DEFER_WGRAD and MATH != "f32" and live and live[0].is_cuda):
        return False
    for p in live:
        if not (p.is_leaf and p.requires_grad and p._base is None) or not _hooks_allow_side_stream(p):
            return False
        if getattr(p, "_post_accumulate_grad_hooks", None) and _FORWARD_USES.get(id(p), 1) > 1:
            return False
    if not _JOIN_QUEUED[0]:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_final_flush)
        except RuntimeError:
            return False
        _JOIN_QUEUED[0] = True
    device = live[0].device
    side = _wgrad_side_stream(device)
    side.wait_stream(torch.cuda.current_stream(device))
    for b in operands:
        if b is not None:
            b.record_stream(side)
    with torch.no_grad(), torch.cuda.stream(side):
        grads = compute()
        for p, g in zip(params, grads):
            if p is None or g is None:
                continue
            g = g.view(p.shape) if g.shape != p.shape else g
            if getattr(p, "_post_accumulate_grad_hooks", None):
                _deliver_grad(p, g)        # (the exchange's hook runs here, under the side stream)
            elif p.grad is None:
                p.grad = g
            else:
                p.grad.add_(g)
    return True


@torch.no_grad()
def flush_deferred_weight_gradients():
    global _DEFERRED
    if not _DEFERRED:
        return
    if WGRAD_STREAM and not _ON_SIDE_STREAM[0] and \
            all(_hooks_allow_side_stream(it[2]) for it in _DEFERRED) and \
            getattr(getattr(_DEFERRED[0][1], "buf", None), "is_cuda", False):
        device = _DEFERRED[0][1].buf.device
        side = _wgrad_side_stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        for g, x, *_ in _DEFERRED:        # operand planes were allocated on the compute stream: keep them until the side stream is done
            g.buf.record_stream(side)
            x.buf.record_stream(side)
        _ON_SIDE_STREAM[0] = True
        try:
            with torch.cuda.stream(side):
                flush_deferred_weight_gradients()
        finally:
            _ON_SIDE_STREAM[0] = False
        return
    items, _DEFERRED = _DEFERRED, []
    groups = {}
    for it in items:
        g, x, w, stride, pad, dil, _ = it
        groups.setdefault((tuple(x.shape), tuple(w.shape), stride, pad, dil), []).append(it)
    lib = L.lib()
    for (x_shape, w_shape, stride, pad, dil), members in groups.items():
        pl = _plan(x_shape, w_shape, stride, pad, dil)
        set_segment(_DEFERRED_SEGMENT.get(id(members[0][2]), SEGMENT))
        for i0 in range(0, len(members), GROUP_MAX):
            part = members[i0:i0 + GROUP_MAX]
            if len(part) == 1 or not pl.x3[2]:
                for g, x, w, _, _, _, rs in part:
                    _deliver_grad(w, planes_backward_weight(g, x, w, stride, pad, dil, row_scale=rs))
                continue
            n = len(part)
            outs = []
            for g, x, w, _, _, _, rs in part:
                slot = grad_slot(w)
                taps = w_shape[2] * w_shape[3]
                if slot is not None:
                    slot = slot.as_strided(w_shape, (taps * w_shape[1], 1, w_shape[3] * w_shape[1], w_shape[1]))
                outs.append(slot if slot is not None else
                            torch.empty(w_shape, dtype=torch.float32, device=x.device, memory_format=CL))
            ptr_t = C.c_void_p * n
            gh, gl = zip(*[_hl(g.buf) for g, *_ in part])
            xh, xl = zip(*[_hl(x.buf) for _, x, *_ in part])
            val = lambda v: v.value if isinstance(v, C.c_void_p) else v   # noqa: E731
            dws = ptr_t(*[o.data_ptr() for o in outs])
            scales = ptr_t(*[(it[6].data_ptr() if it[6] is not None else None) for it in part])
            nbytes = lib.jtsm_conv_bf16x3_wgrad_group_workspace_bytes(pl.ref, n)
            ws = _scratch(nbytes, outs[0].device)
            variant = None
            if LAUNCH_LOG is not None:
                tile, sp = C.c_int(), C.c_int()
                L.check(lib.jtsm_conv_bf16x3_wgrad_group_plan(pl.ref, n, C.byref(tile), C.byref(sp)), "wgrad_group_plan")
                np_ = ",1" if MATH == "f16" else ""
                if tile.value == 0:
                    base = "igemm_x3_wgrad_halo_group_kernel" + ("<1>" if MATH == "f16" else "")
                else:
                    base = "igemm_x3_wgrad_group_kernel<%s,2%s>" % ("4,2,2,4" if tile.value == 256 else "2,2,2,2", np_)
                variant = _Variant(base, sp.value)
            desc = pl.desc[:-1] + (pl.desc[-1] * n,) if pl.desc is not None else None
            if MATH == "f16":
                call = lambda: lib.jtsm_conv2d_backward_weight_group_f16(      # noqa: E731
                    n, ptr_t(*[val(v) for v in gh]), ptr_t(*[val(v) for v in xh]), dws, scales, pl.ref, GRAD_SHIFT,
                    L.ptr(ws), C.c_size_t(nbytes), L.stream())
            else:
                call = lambda: lib.jtsm_conv2d_backward_weight_group_bf16x3(   # noqa: E731
                    n, ptr_t(*[val(v) for v in gh]), ptr_t(*[val(v) for v in gl]), ptr_t(*[val(v) for v in xh]),
                    ptr_t(*[val(v) for v in xl]), dws, scales, pl.ref, L.ptr(ws), C.c_size_t(nbytes), L.stream())
            L.check(_timed(variant, pl.flops * n, call, desc, 0, outs[0].numel() * n), "conv2d_backward_weight_group")
            for (g, x, w, *_), o in zip(part, outs):
                _deliver_grad(w, o)


def planes_channel_sum(g, grad=True):
    """Bias gradient of a gradient held as planes: sum over every axis but channels."""
    ch = g.shape[1]
    rows = g.numel // ch
    out = torch.empty(ch, dtype=torch.float32, device=g.device)
    nbytes = 4 * 1024 * ch
    ws = _scratch(nbytes, g.device)
    gh, gl = _hl(g.buf)
    L.note_bytes((2.0 if MATH == "f16" else 4.0) * g.numel)
    L.check(L.lib().jtsm_channel_sum_planes(gh, gl, L.ptr(out), C.c_long(rows), ch,
                                                  GRAD_SHIFT if (grad and MATH == "f16") else 0, L.ptr(ws),
                                                  C.c_size_t(nbytes), L.stream()), "channel_sum_planes")
    return out


def planes_channel_sum_multi(gs, grad=True):
    """planes_channel_sum of up to 8 gradients of the same width in one launch (+ one fold) -> list of (C,) tensors."""
    if not gs:
        return []
    ch = gs[0].shape[1]
    assert all(g.shape[1] == ch for g in gs) and len(gs) <= 8
    n = len(gs)
    outs = torch.empty((n, ch), dtype=torch.float32, device=gs[0].device)
    nbytes = 4 * 1024 * ch * n
    ws = _scratch(nbytes, gs[0].device)
    his = (C.c_void_p * n)(*[g.buf.data_ptr() for g in gs])
    los = None if MATH == "f16" else (C.c_void_p * n)(*[g.buf.data_ptr() + g.buf.numel() for g in gs])
    optr = (C.c_void_p * n)(*[outs[i].data_ptr() for i in range(n)])
    rows = (C.c_long * n)(*[g.numel // ch for g in gs])
    L.note_bytes((2.0 if MATH == "f16" else 4.0) * sum(g.numel for g in gs))
    L.check(L.lib().jtsm_channel_sum_planes_multi(his, los, optr, rows, n, ch,
                                                  GRAD_SHIFT if (grad and MATH == "f16") else 0, L.ptr(ws),
                                                  C.c_size_t(nbytes), L.stream()), "channel_sum_planes_multi")
    return [outs[i] for i in range(n)]


def planes_conv_transpose2x2_forward(x, w, bias=None, relu=False, fp32=False):
    """conv_transpose2x2_forward on a PlaneTensor -> PlaneTensor of y (or the fp32 y with fp32=True)."""
    n, i, h, wd = x.shape
    o = w.shape[1]
    oshape = (n, o, 2 * h, 2 * wd)
    y = torch.empty(oshape, dtype=torch.float32, device=x.device, memory_format=CL) if fp32 else None
    yp = PlaneTensor.empty(oshape, x.device)
    xh, xl = _hl(x.buf)
    w1 = w.as_strided((i, 4 * o, 1, 1), (4 * o, 1, 1, 1))
    wth, wtl = _hl(_weight_planes(w1, True))
    yh, yl = _hl(yp.buf)
    lib = L.lib()
    flops = 2.0 * n * h * wd * i * 4 * o
    var = _x3_variant(_plan((n, i, h, wd), (4 * o, i, 1, 1), 1, 0, 1).s, 0)
    var = _Variant(str(var), 1) if var is not None else None
    if MATH == "f16":
        L.check(_timed(var, flops, lambda: lib.jtsm_conv_transpose2x2_forward_f16(
            xh, wth, L.ptr(y), yh, n, h, wd, i, o, L.ptr(bias), int(bool(relu)), L.stream()),
            _ct_desc(n, h, wd, i, o), 0, yp.numel, True, fp32), "conv_transpose2x2_forward_f16")
    else:
        L.check(_timed(var, flops, lambda: lib.jtsm_conv_transpose2x2_forward_bf16x3(
            xh, xl, wth, wtl, L.ptr(y), yh, yl, n, h, wd, i, o, L.ptr(bias), int(bool(relu)), L.stream()),
            _ct_desc(n, h, wd, i, o), 0, yp.numel, True, fp32), "conv_transpose2x2_forward_bf16x3")
    if y is not None:
        planes_put(y, yp.buf)
    return (y, yp) if fp32 else yp


def planes_conv_transpose2x2_backward_data(g, w, gate=None, bias_out=None):
    """PlaneTensor g (N, out, 2H, 2W) -> PlaneTensor of dx (N, in, H, W), gated by the PlaneTensor `gate`.
    bias_out (in,): see planes_backward_data — then returns (dx, whether bias_out was written)."""
    n, o, h2, w2 = g.shape
    i = w.shape[0]
    h, wd = h2 // 2, w2 // 2
    pl = _plan(g.shape, (i, o, 2, 2), 2, 0, 1)
    dp = PlaneTensor.empty((n, i, h, wd), g.device)
    part = _colsum_partials(pl, 0, i, g.device) if bias_out is not None else None
    if part is not None:
        gh, gl = _hl(g.buf)
        wh, wl = _hl(_weight_planes(w))
        dh, dl = _hl(dp.buf)
        gate_h = _hl(gate.buf)[0] if gate is not None else None
        nbytes = pl.ws[0]
        ws = _scratch(nbytes, g.device)
        lib = L.lib()
        extra = 0.5 * dp.numel if gate is not None else 0
        if MATH == "f16":
            L.check(_timed(_x3_variant(pl.s, 0), pl.flops, lambda: lib.jtsm_conv_transpose2x2_backward_data_colsum_f16(
                gh, wh, None, dh, n, h, wd, i, o, None, gate_h, GRAD_SHIFT, L.ptr(part), L.ptr(ws), C.c_size_t(nbytes),
                L.stream()), pl.desc, extra, dp.numel, True, False), "conv_transpose2x2_backward_data_colsum_f16")
        else:
            L.check(_timed(_x3_variant(pl.s, 0), pl.flops, lambda: lib.jtsm_conv_transpose2x2_backward_data_colsum_bf16x3(
                gh, gl, wh, wl, None, dh, dl, n, h, wd, i, o, None, gate_h, L.ptr(part), L.ptr(ws), C.c_size_t(nbytes),
                L.stream()), pl.desc, extra, dp.numel, True, False), "conv_transpose2x2_backward_data_colsum_bf16x3")
        bias_out.take(part)
        return dp, True
    gh, gl = _hl(g.buf)
    wh, wl = _hl(_weight_planes(w))
    dh, dl = _hl(dp.buf)
    gate_h = _hl(gate.buf)[0] if gate is not None else None
    nbytes = pl.ws[0]
    ws = _scratch(nbytes, g.device)
    lib = L.lib()
    extra = 0.5 * dp.numel if gate is not None else 0
    if MATH == "f16":
        L.check(_timed(_x3_variant(pl.s, 0), pl.flops, lambda: lib.jtsm_conv_transpose2x2_backward_data_f16(
            gh, wh, None, dh, n, h, wd, i, o, None, gate_h, GRAD_SHIFT, L.ptr(ws), C.c_size_t(nbytes), L.stream()),
            pl.desc, extra, dp.numel, True, False), "conv_transpose2x2_backward_data_f16")
    else:
        L.check(_timed(_x3_variant(pl.s, 0), pl.flops, lambda: lib.jtsm_conv_transpose2x2_backward_data_bf16x3(
            gh, gl, wh, wl, None, dh, dl, n, h, wd, i, o, None, gate_h, L.ptr(ws), C.c_size_t(nbytes), L.stream()),
            pl.desc, extra, dp.numel, True, False), "conv_transpose2x2_backward_data_bf16x3")
    return (dp, False) if bias_out is not None else dp


def planes_conv_transpose2x2_backward_weight(g, x, w):
    """dW of the transposed convolution from PlaneTensors: the weight gradient of the 2x2 / stride-2 convolution it
    transposes, with that convolution's 'dy' = x and input = g."""
    i, o = w.shape[0], w.shape[1]
    return planes_backward_weight(x, g, w, 2, 0, 1, w_shape=(i, o, 2, 2))


class _ConvTranspose2x2(Function):
    @staticmethod
    def forward(ctx, x, w, bias, relu):
        y = conv_transpose2x2_forward(x, w, bias, relu, emit_planes=True)
        ctx.relu = relu
        ctx.save_for_backward(x, w, y if relu else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        from .elementwise import channel_sum, relu_backward
        x, w, y = ctx.saved_tensors
        g = relu_backward(dy, y, emit_planes=True) if ctx.relu else _cl(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = conv_transpose2x2_backward_data(g, w)
        if ctx.needs_input_grad[1]:
            dw = conv_transpose2x2_backward_weight(g, x, w)
        if ctx.needs_input_grad[2]:
            db = channel_sum(g)
        return dx, dw, db, None


def conv_transpose2x2_fused(x, w, bias=None, relu=False):
    return _ConvTranspose2x2.apply(x, w, bias, relu)


def _wgrad_bias_fits(pl):
    """The bias gradient rides in the weight-gradient contraction on the 128x128-tile and 3x3-halo kernels; on the
    256x256-tile kernel its extra accumulators cost the whole launch ~20 % (measured: 144 -> 173 us on the mask heads'
    3x3 layers, 615 -> 760 us on fc1), more than a separate sum over the gradient's planes."""
    big = getattr(pl, "wgrad_big", None)
    if big is None:
        v = [C.c_int() for _ in range(6)]
        L.check(L.lib().jtsm_conv_bf16x3_plan(pl.ref, 2, *[C.byref(x) for x in v]), "conv_bf16x3_plan")
        big = pl.wgrad_big = (v[0].value == 4 and v[4].value != 0)      # WM == 4: the 256x256 tile (nbuf 0 = halo)
    return not big


def wgrad_bias_fits(x_shape, w_shape, stride=1, pad=0, dil=1):
    """Will (conv2d|planes)_backward_weight(..., bias_out=) compute the bias gradient inside the contraction?  (Callers
    that can batch the separate sums — the mask tower — ask first.)"""
    if MATH == "f32" or not BIAS_IN_WGRAD:
        return False
    pl = _plan(x_shape, w_shape, stride, pad, dil)
    return bool(pl.x3[2]) and _wgrad_bias_fits(pl)


def _wgrad_bias_call(pl, gh, gl, xh, xl, out, bias_out, row_scale, fresh, device):
    """Weight gradient + bias gradient in one contraction (jtsm_conv2d_backward_weight_bias_*)."""
    nbytes = max(pl.ws[3], pl.ws[2])
    ws = _scratch(nbytes, device)
    lib = L.lib()
    if MATH == "f16":
        L.check(_timed(_x3_variant(pl.s, 2), pl.flops, lambda: lib.jtsm_conv2d_backward_weight_bias_f16(
            gh, xh, L.ptr(out), L.ptr(bias_out), pl.ref, L.ptr(row_scale), int(fresh), GRAD_SHIFT, L.ptr(ws),
            C.c_size_t(nbytes), L.stream()), pl.desc, 0, out.numel()), "conv2d_backward_weight_bias_f16")
    else:
        L.check(_timed(_x3_variant(pl.s, 2), pl.flops, lambda: lib.jtsm_conv2d_backward_weight_bias_bf16x3(
            gh, gl, xh, xl, L.ptr(out), L.ptr(bias_out), pl.ref, L.ptr(row_scale), int(fresh), L.ptr(ws),
            C.c_size_t(nbytes), L.stream()), pl.desc, 0, out.numel()), "conv2d_backward_weight_bias_bf16x3")


class _ConvFused(Function):
    """y = relu?(conv(x, w) * scale + bias + residual); scale/bias are constants of the op
    (FrozenBN statistics or a conv bias treated by the caller), residual gets dy * relu'."""

    @staticmethod
    def forward(ctx, x, w, scale, bias, residual, stride, pad, dil, relu, bias_needs_grad, emit_planes=True,
                emit_dx_planes=False, fan=None):
        ctx.segment = SEGMENT
        ctx.bias_param = bias if bias_needs_grad else None
        if w.requires_grad:
            _FORWARD_USES[id(w)] = _FORWARD_USES.get(id(w), 0) + 1
            if ctx.bias_param is not None:
                _FORWARD_USES[id(bias)] = _FORWARD_USES.get(id(bias), 0) + 1
        y = conv2d_forward(x, w, stride, pad, dil, scale, bias, residual, relu, emit_planes=emit_planes)
        ctx.fan = fan   # (layers/grad_fan.py: conv2d_fused claimed the input's fan view, if it is one)
        ctx.emit_dx_planes = emit_dx_planes   # the input's gradient is the dy of another contraction (FPN laterals)
        ctx.cfg = (stride, pad, dil, relu, bias_needs_grad, tuple(x.shape), tuple(w.shape))
        ctx.has_res = residual is not None
        ctx.save_for_backward(x, w, scale, y if relu else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        from .elementwise import channel_sum, relu_backward

        set_segment(ctx.segment)
        if ctx.segment == "fpn":
            _heads_done()
        x, w, scale, y = ctx.saved_tensors
        stride, pad, dil, relu, bias_needs_grad, xs, ws = ctx.cfg
        x3 = MATH != "f32" and dy.shape[0] > 0 and dy.shape[1] % 8 == 0 and \
            (ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
        if relu:
            g = relu_backward(dy, y, emit_planes=x3)   # one pass: gate, and the planes both gradients contract
        else:
            g = _cl(dy)
        dx = dw = db = dres = None
        # (Launching dw on a second HIP stream beside dx was measured on MI355X: 46.6 vs 45.6 ms per step —
        # slower; the contractions already hold the chip at its power-limited clock.  Kept in order.)
        if ctx.needs_input_grad[0]:
            # fold the FrozenBN scale into the weight rows once (a few MB) so the data-gradient GEMM takes
            # the direct-to-LDS path, which cannot rescale operands on the fly
            # another consumer of x wrote its gradient map already this backward pass: add into it (layers/grad_fan.py)
            # (a gradient that other consumers add to is not final: its planes are nobody's operand)
            sink = grad_fan.target(ctx.fan, xs, g.device)
            if MATH != "f32":   # (ineligible shapes fall through to the fp32 kernel's own kscale path)
                dx = conv2d_backward_data(g, w, xs, stride, pad, dil, kscale=scale,
                                          emit_planes=ctx.emit_dx_planes and ctx.fan is None, into=sink)
            else:
                w_eff = w if scale is None else (w * scale.view(-1, 1, 1, 1)).contiguous(memory_format=CL)
                dx = conv2d_backward_data(g, w_eff, xs, stride, pad, dil, into=sink)
            if sink is not None or grad_fan.offer(ctx.fan, dx):
                dx = None      # (in the fan's record: autograd gets it from the fan node)
        want_db = bias_needs_grad and ctx.needs_input_grad[3]
        if ctx.needs_input_grad[1]:
            def run():
                db_ = torch.empty(ws[0], dtype=g.dtype, device=g.device) if want_db else None   # (rides in the contraction)
                dw_ = conv2d_backward_weight(g, x, ws, stride, pad, dil, row_scale=scale, w=w, bias_out=db_)
                if dw_.stride() != w.stride() and w.shape[2] == 1 and w.shape[3] == 1:
                    dw_ = dw_.as_strided(w.shape, w.stride())
                return dw_, db_
            # beside the rest of the backward on the weight-gradient side stream when the parameters are plain leaves
            # (the operands' planes live in the step's plane cache; g and x themselves may be freed behind this node)
            if x3 and _plan(x.shape, ws, stride, pad, dil).x3[2] and \
                    side_weight_gradients([w, ctx.bias_param if want_db else None], run, (g, x)):
                dw = db = None
            else:
                dw, db = run()
        elif want_db:
            db = channel_sum(g)
        if dw is not None and dw.stride() != w.stride() and w.shape[2] == 1 and w.shape[3] == 1:
            # 1x1 weights: OHWI and OIHW are the same bytes; hand the gradient back with the parameter's own
            # strides (DDP bucket views and the fused SGD compare strides, not byte order)
            dw = dw.as_strided(w.shape, w.stride())
        if ctx.has_res and ctx.needs_input_grad[4]:
            dres = g
        return dx, dw, None, db, dres, None, None, None, None, None, None, None, None


ZERO_PADDED = "_jtsm_zero_padded_channels"   # set on a gradient buffer whose trailing (padding) channels are zero


class _LeadingChannels(Function):
    """y[:, :o] of a channel-padded result.  Autograd's own slice hands back zeros(y.shape) with the gradient copied
    into it, in NCHW order, which the convolution's backward then copies once more into channels-last: three passes
    over the widest map of the semantic head (0.06 ms per step).  A consumer that computed its gradient in a buffer of
    the PADDED width with zero padding (the fused up-sample + cross-entropy does, and says so with ZERO_PADDED) gets
    that buffer passed through instead."""

    passed_through = 0     # (tests: how many backward calls took the zero-padded buffer as it was)

    @staticmethod
    def forward(ctx, y, o):
        ctx.full = y.shape[1]
        return y[:, :o]

    @staticmethod
    def backward(ctx, g):
        base = g._base
        if (base is not None and getattr(base, ZERO_PADDED, False) and base.dim() == 4 and
                base.shape[3] == ctx.full and base.is_contiguous() and g.data_ptr() == base.data_ptr() and
                tuple(g.shape) == (base.shape[0], g.shape[1], base.shape[1], base.shape[2]) and
                g.stride() == (base.stride(0), 1, base.stride(1), base.stride(2))):
            _LeadingChannels.passed_through += 1
            return base.permute(0, 3, 1, 2), None
        return torch.nn.functional.pad(g, (0, 0, 0, 0, 0, ctx.full - g.shape[1])), None


def conv2d_fused(x, w, scale=None, bias=None, residual=None, stride=1, pad=0, dil=1, relu=False,
                 bias_needs_grad=False, emit_planes=True, emit_dx_planes=False):
    """Autograd-aware fused convolution.  An output-channel count that is not a multiple of 4 (54 sem-seg
    classes, the 1870-wide fused predictor) is zero-padded up for the kernels' 16-byte rows and the
    padding is sliced off the result (its gradient is zero by construction)."""
    fan = grad_fan.claim(x)
    o = w.shape[0]
    if o % 4:
        extra = 4 - o % 4
        w = torch.cat([w, w.new_zeros((extra,) + tuple(w.shape[1:]))]).contiguous(memory_format=CL)
        if scale is not None:
            scale = torch.cat([scale, scale.new_ones(extra)])
        if bias is not None:
            bias = torch.cat([bias, bias.new_zeros(extra)])
        if residual is not None:
            residual = torch.nn.functional.pad(residual, (0, 0, 0, 0, 0, extra))
        y = _ConvFused.apply(x, w, scale, bias, residual, stride, pad, dil, relu, bias_needs_grad, False, False, fan)
        return _LeadingChannels.apply(y, o)
    return _ConvFused.apply(x, w, scale, bias, residual, stride, pad, dil, relu, bias_needs_grad, emit_planes,
                            emit_dx_planes, fan)


def linear_fused(x, w, bias=None, relu=False, bias_needs_grad=True):
    """x (R, in) @ w(out, in)^T + bias, optional ReLU — the 1x1 case of conv2d_fused."""
    r, k = x.shape
    y = conv2d_fused(x.view(r, k, 1, 1), w.view(w.shape[0], k, 1, 1), None, bias, None, 1, 0, 1, relu,
                     bias_needs_grad)
    return y.view(r, w.shape[0])


class _ColumnSplit(Function):
    """Views of consecutive column ranges of a (R, C) matrix whose gradients come back as ONE concatenation —
    instead of autograd's zero-filled (R, C) buffer plus an add per slice."""

    @staticmethod
    def forward(ctx, y, *sizes):
        ctx.sizes, ctx.rows = sizes, y.shape[0]
        outs, c0 = [], 0
        for n in sizes:
            outs.append(y[:, c0:c0 + n])
            c0 += n
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        ref = next(g for g in grads if g is not None)
        parts = [g if g is not None else ref.new_zeros((ctx.rows, n)) for g, n in zip(grads, ctx.sizes)]
        return (torch.cat(parts, dim=1),) + (None,) * len(ctx.sizes)


def linear_fused_split(x, weights, biases, relu=False):
    """Several Linear layers on the same input as ONE GEMM (weights concatenated along `out`); returns one output
    view per layer.  The column count is padded to a multiple of 4 for the kernels' 16-byte rows."""
    r, k = x.shape
    sizes = [w.shape[0] for w in weights]
    total = sum(sizes)
    pad = (-total) % 4
    ws, bs = list(weights), list(biases)
    if pad:
        ws.append(weights[0].new_zeros((pad, k)))
        bs.append(biases[0].new_zeros(pad))
    w = torch.cat(ws).view(total + pad, k, 1, 1)
    b = torch.cat(bs)
    y = _ConvFused.apply(x.view(r, k, 1, 1), w, None, b, None, 1, 0, 1, relu, True, False, False).view(r, total + pad)
    outs = _ColumnSplit.apply(y, *(sizes + ([pad] if pad else [])))
    return list(outs[:len(sizes)])
