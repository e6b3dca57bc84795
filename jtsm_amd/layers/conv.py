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

CL = torch.channels_last

# Optional per-launch timing (bench.py's roofline leg): when a list, every contraction launch appends
# (kernel variant, algorithmic FLOPs, start event, stop event), recorded on the launch stream.
LAUNCH_LOG = None


# Contraction arithmetic: "f32" = exact fp32 MFMA; "bf16x3" = split-bf16 (three bf16 MFMA products per fp32
# product, fp32 accumulate, ~2^-16 relative error per product) for every eligible forward / data-gradient
# contraction.  See csrc/conv_x3.h.
MATH = os.environ.get("JTSM_CONV_MATH", "bf16x3")


def set_math(mode):
    global MATH
    if mode not in ("f32", "bf16x3"):
        raise ValueError("conv math must be 'f32' or 'bf16x3', got %r" % (mode,))
    MATH = mode


def split_bf16(t):
    """(hi, lo) bf16 planes (int16 bit patterns) of a float32 tensor, in its storage order."""
    flat = t.permute(0, 2, 3, 1) if t.dim() == 4 else t
    if not flat.is_contiguous():
        raise RuntimeError("split_bf16: tensor must be channels_last (4-d) or contiguous")
    hi = torch.empty(t.numel(), dtype=torch.int16, device=t.device)
    lo = torch.empty(t.numel(), dtype=torch.int16, device=t.device)
    L.check(L.lib().jtsm_split_bf16_f32(L.ptr(t), L.ptr(hi), L.ptr(lo), C.c_long(t.numel()), L.stream()),
            "split_bf16")
    return hi, lo


def split_bf16_transposed(w, row_scale=None):
    """Planes of W^T [in][taps][out] of a channels_last (out, in, kh, kw) weight, rows pre-multiplied by
    row_scale[out] when given."""
    o, i, kh, kw = w.shape
    hi = torch.empty(w.numel(), dtype=torch.int16, device=w.device)
    lo = torch.empty(w.numel(), dtype=torch.int16, device=w.device)
    L.check(L.lib().jtsm_split_bf16_transposed_f32(L.ptr(w), L.ptr(row_scale), L.ptr(hi), L.ptr(lo), o, kh * kw, i,
                                                   L.stream()),
            "split_bf16_transposed")
    return hi, lo


# bf16 planes already made this step, keyed by the tensor's memory: a conv epilogue or relu_backward that
# emitted the planes of its output registers them here, and the next contraction that consumes the tensor
# (forward input, weight-gradient input, shared block input of conv1 + shortcut + FPN lateral) finds them.
# Entries hold the tensor (detached), so its memory cannot be recycled under a live key; the model clears
# the cache at the start of every forward (planes_clear).
_PLANES = {}
_PLANES_MAX = 4096


def planes_clear():
    _PLANES.clear()


def _pkey(t):
    # planes mirror the flat memory of a DENSE tensor, so any dense view of the same bytes shares them
    return (t.data_ptr(), t.numel())


def planes_put(t, hi, lo):
    if len(_PLANES) >= _PLANES_MAX:
        _PLANES.clear()
    _PLANES[_pkey(t)] = (t.detach(), t._version, hi, lo)


def planes_of(t):
    """Cached (hi, lo) planes of a tensor, splitting it now if nobody has."""
    e = _PLANES.get(_pkey(t))
    if e is not None and e[1] == t._version:
        return e[2], e[3]
    hi, lo = split_bf16(t)
    planes_put(t, hi, lo)
    return hi, lo


def _x3(s, role):
    return MATH == "bf16x3" and s.batch > 0 and bool(L.lib().jtsm_conv_bf16x3_eligible(C.byref(s), role))


def _x3_variant(s, role):
    n = s.out_c if role == 0 else s.in_c
    return "igemm_x3_kernel<%s,%s,2>" % (_ROLE_NAME[role], "256,64" if n <= 64 else "128,128")


def _timed(variant, flops, call, shape=None):
    if LAUNCH_LOG is None:
        return call()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    rc = call()
    b.record()
    LAUNCH_LOG.append((variant, flops, a, b, shape))
    return rc


_ROLE_NAME = ("FWD", "DGRAD", "WGRAD")


def _variant(s, role, has_kscale=False):
    """Exact kernel instantiation the library will launch for this call (mirrors its dispatch)."""
    if LAUNCH_LOG is None:
        return None
    k, bm, bn, sp = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    L.check(L.lib().jtsm_conv_plan(C.byref(s), role, int(has_kscale), C.byref(k), C.byref(bm), C.byref(bn),
                                   C.byref(sp)), "conv_plan")
    if k.value == 0:
        return "igemm_kernel<%s,%d,%d>" % (_ROLE_NAME[role], bm.value, bn.value)
    return "igemm_dma_kernel<%s,%d,%d,%d>" % (_ROLE_NAME[role], bm.value, bn.value, 2 if k.value == 1 else 1)


def _desc(s):
    return (s.batch, s.in_h, s.in_w, s.in_c, s.out_c, s.kernel_h, s.stride)


def _workspace(s, backward_data, device):
    """Split-K scratch for this shape (None when the layer is not split)."""
    nbytes = L.lib().jtsm_conv_workspace_bytes(C.byref(s), backward_data)
    if nbytes == 0:
        return None, 0
    return torch.empty(nbytes, dtype=torch.uint8, device=device), nbytes


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
    s = _shape(x.shape, w.shape, stride, pad, dil)
    oh, ow = out_hw(s)
    y = torch.empty((s.batch, s.out_c, oh, ow), dtype=x.dtype, device=x.device, memory_format=CL)
    if residual is not None:
        residual = _cl(residual)
        assert residual.shape == y.shape
    ws, nbytes = _workspace(s, 0, x.device)
    if _x3(s, 0):
        xh, xl = planes_of(x)
        wh, wl = split_bf16(w)
        yh = yl = None
        if emit_planes and s.out_c % 8 == 0:
            yh = torch.empty(y.numel(), dtype=torch.int16, device=y.device)
            yl = torch.empty(y.numel(), dtype=torch.int16, device=y.device)
        L.check(_timed(_x3_variant(s, 0), _flops(s), lambda: L.lib().jtsm_conv2d_forward_bf16x3(
            L.ptr(xh), L.ptr(xl), L.ptr(wh), L.ptr(wl), L.ptr(y), L.ptr(yh), L.ptr(yl), C.byref(s), L.ptr(scale),
            L.ptr(bias), L.ptr(residual), int(bool(relu)), L.ptr(ws), C.c_size_t(nbytes), L.stream()), _desc(s)),
                "conv2d_forward_bf16x3")
        if yh is not None:
            planes_put(y, yh, yl)
        return y
    variant = _variant(s, 0)
    L.check(_timed(variant, _flops(s), lambda: L.lib().jtsm_conv2d_forward_f32(
        L.ptr(x), L.ptr(w), L.ptr(y), C.byref(s), L.ptr(scale), L.ptr(bias), L.ptr(residual), int(bool(relu)),
        L.ptr(ws), C.c_size_t(nbytes), L.stream()), _desc(s)), "conv2d_forward")
    return y


def conv2d_backward_data(dy, w, x_shape, stride=1, pad=0, dil=1, kscale=None, accumulate=None,
                         relu_mask=None):
    _check(dy, w, kscale, accumulate, relu_mask)
    dy, w = _cl(dy), _cl(w)
    s = _shape(x_shape, w.shape, stride, pad, dil)
    dx = torch.empty(tuple(x_shape), dtype=dy.dtype, device=dy.device, memory_format=CL)
    if accumulate is not None:
        accumulate = _cl(accumulate)
    if relu_mask is not None:
        relu_mask = _cl(relu_mask)
    ws, nbytes = _workspace(s, 1, dy.device)
    if _x3(s, 1):
        gh, gl = planes_of(dy)
        wh, wl = split_bf16_transposed(w, kscale)   # the per-row scale rides along in the transposing split
        L.check(_timed(_x3_variant(s, 1), _flops(s), lambda: L.lib().jtsm_conv2d_backward_data_bf16x3(
            L.ptr(gh), L.ptr(gl), L.ptr(wh), L.ptr(wl), L.ptr(dx), C.byref(s), L.ptr(accumulate), L.ptr(relu_mask),
            L.ptr(ws), C.c_size_t(nbytes), L.stream()), _desc(s)), "conv2d_backward_data_bf16x3")
        return dx
    variant = _variant(s, 1, kscale is not None)
    L.check(_timed(variant, _flops(s), lambda: L.lib().jtsm_conv2d_backward_data_f32(
        L.ptr(dy), L.ptr(w), L.ptr(dx), C.byref(s), L.ptr(kscale), L.ptr(accumulate), L.ptr(relu_mask),
        L.ptr(ws), C.c_size_t(nbytes), L.stream()), _desc(s)), "conv2d_backward_data")
    return dx


def conv2d_backward_weight(dy, x, w_shape, stride=1, pad=0, dil=1, row_scale=None, out=None):
    _check(dy, x, row_scale)
    dy, x = _cl(dy), _cl(x)
    s = _shape(x.shape, w_shape, stride, pad, dil)
    if _x3(s, 2):
        gh, gl = planes_of(dy)
        xh, xl = planes_of(x)
        fresh = out is None
        if fresh:   # deterministic slab kernel: writes every element, nothing to clear
            out = torch.empty(tuple(w_shape), dtype=x.dtype, device=x.device, memory_format=CL)
        nbytes = L.lib().jtsm_conv_bf16x3_wgrad_workspace_bytes(C.byref(s))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes else None
        L.check(_timed("igemm_x3_wgrad_kernel<2>", _flops(s), lambda: L.lib().jtsm_conv2d_backward_weight_bf16x3(
            L.ptr(gh), L.ptr(gl), L.ptr(xh), L.ptr(xl), L.ptr(out), C.byref(s), L.ptr(row_scale), int(fresh),
            L.ptr(ws), C.c_size_t(nbytes), L.stream()), _desc(s)), "conv2d_backward_weight_bf16x3")
        return out
    zero = False   # cleared here (not inside the timed launch) so per-launch timings are kernel-only
    if out is None:
        out = torch.empty(tuple(w_shape), dtype=x.dtype, device=x.device, memory_format=CL).zero_()
    L.check(_timed(_variant(s, 2), _flops(s), lambda: L.lib().jtsm_conv2d_backward_weight_f32(
        L.ptr(dy), L.ptr(x), L.ptr(out), C.byref(s), L.ptr(row_scale), int(zero), L.stream()), _desc(s)),
            "conv2d_backward_weight")
    return out


class _ConvFused(Function):
    """y = relu?(conv(x, w) * scale + bias + residual); scale/bias are constants of the op
    (FrozenBN statistics or a conv bias treated by the caller), residual gets dy * relu'."""

    @staticmethod
    def forward(ctx, x, w, scale, bias, residual, stride, pad, dil, relu, bias_needs_grad, emit_planes=True):
        y = conv2d_forward(x, w, stride, pad, dil, scale, bias, residual, relu, emit_planes=emit_planes)
        ctx.cfg = (stride, pad, dil, relu, bias_needs_grad, tuple(x.shape), tuple(w.shape))
        ctx.has_res = residual is not None
        ctx.save_for_backward(x, w, scale, y if relu else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        from .elementwise import channel_sum, relu_backward

        x, w, scale, y = ctx.saved_tensors
        stride, pad, dil, relu, bias_needs_grad, xs, ws = ctx.cfg
        x3 = MATH == "bf16x3" and dy.shape[0] > 0 and dy.shape[1] % 8 == 0 and \
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
            if MATH == "bf16x3":   # (ineligible shapes fall through to the fp32 kernel's own kscale path)
                dx = conv2d_backward_data(g, w, xs, stride, pad, dil, kscale=scale)
            else:
                w_eff = w if scale is None else (w * scale.view(-1, 1, 1, 1)).contiguous(memory_format=CL)
                dx = conv2d_backward_data(g, w_eff, xs, stride, pad, dil)
        if ctx.needs_input_grad[1]:
            dw = conv2d_backward_weight(g, x, ws, stride, pad, dil, row_scale=scale)
        if bias_needs_grad and ctx.needs_input_grad[3]:
            db = channel_sum(g)
        if ctx.has_res and ctx.needs_input_grad[4]:
            dres = g
        return dx, dw, None, db, dres, None, None, None, None, None, None


def conv2d_fused(x, w, scale=None, bias=None, residual=None, stride=1, pad=0, dil=1, relu=False,
                 bias_needs_grad=False, emit_planes=True):
    """Autograd-aware fused convolution.  An output-channel count that is not a multiple of 4 (54 sem-seg
    classes, the 1870-wide fused predictor) is zero-padded up for the kernels' 16-byte rows and the
    padding is sliced off the result (its gradient is zero by construction)."""
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
        y = _ConvFused.apply(x, w, scale, bias, residual, stride, pad, dil, relu, bias_needs_grad, False)
        return y[:, :o]
    return _ConvFused.apply(x, w, scale, bias, residual, stride, pad, dil, relu, bias_needs_grad, emit_planes)


def linear_fused(x, w, bias=None, relu=False, bias_needs_grad=True):
    """x (R, in) @ w(out, in)^T + bias, optional ReLU — the 1x1 case of conv2d_fused."""
    r, k = x.shape
    y = conv2d_fused(x.view(r, k, 1, 1), w.view(w.shape[0], k, 1, 1), None, bias, None, 1, 0, 1, relu,
                     bias_needs_grad)
    return y.view(r, w.shape[0])
