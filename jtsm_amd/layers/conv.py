"""Convolution / linear primitives of the hot path, on libjtsm_hip.so's fp32-MFMA implicit GEMM.

Tensors keep the reference's logical shapes — activations (N,C,H,W), weights (O,I,kh,kw) — but
must be stored channels_last (NHWC / OHWI in memory), which is what the MI355X kernels read.
`conv2d_fused` is the autograd-aware functional form of what the reference runs as
Conv2d -> FrozenBatchNorm2d -> (+shortcut) -> ReLU (detectron2/layers/wrappers.py:62-83,
batch_norm.py:45-66, modeling/backbone/resnet.py:195-211).
"""
import ctypes as C

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib as L

CL = torch.channels_last

# Optional per-launch timing (bench.py's roofline leg): when a list, every contraction launch appends
# (kernel variant, algorithmic FLOPs, start event, stop event), recorded on the launch stream.
LAUNCH_LOG = None


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


def conv2d_forward(x, w, stride=1, pad=0, dil=1, scale=None, bias=None, residual=None, relu=False):
    _check(x, w, scale, bias, residual)
    x, w = _cl(x), _cl(w)
    s = _shape(x.shape, w.shape, stride, pad, dil)
    oh, ow = out_hw(s)
    y = torch.empty((s.batch, s.out_c, oh, ow), dtype=x.dtype, device=x.device, memory_format=CL)
    if residual is not None:
        residual = _cl(residual)
        assert residual.shape == y.shape
    variant = _variant(s, 0)
    ws, nbytes = _workspace(s, 0, x.device)
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
    variant = _variant(s, 1, kscale is not None)
    ws, nbytes = _workspace(s, 1, dy.device)
    L.check(_timed(variant, _flops(s), lambda: L.lib().jtsm_conv2d_backward_data_f32(
        L.ptr(dy), L.ptr(w), L.ptr(dx), C.byref(s), L.ptr(kscale), L.ptr(accumulate), L.ptr(relu_mask),
        L.ptr(ws), C.c_size_t(nbytes), L.stream()), _desc(s)), "conv2d_backward_data")
    return dx


def conv2d_backward_weight(dy, x, w_shape, stride=1, pad=0, dil=1, row_scale=None, out=None):
    _check(dy, x, row_scale)
    dy, x = _cl(dy), _cl(x)
    s = _shape(x.shape, w_shape, stride, pad, dil)
    zero = False   # cleared here (not inside the timed launch) so per-launch timings are kernel-only
    if out is None:
        out = torch.zeros(tuple(w_shape), dtype=x.dtype, device=x.device).contiguous(memory_format=CL)
    L.check(_timed(_variant(s, 2), _flops(s), lambda: L.lib().jtsm_conv2d_backward_weight_f32(
        L.ptr(dy), L.ptr(x), L.ptr(out), C.byref(s), L.ptr(row_scale), int(zero), L.stream()), _desc(s)),
            "conv2d_backward_weight")
    return out


class _ConvFused(Function):
    """y = relu?(conv(x, w) * scale + bias + residual); scale/bias are constants of the op
    (FrozenBN statistics or a conv bias treated by the caller), residual gets dy * relu'."""

    @staticmethod
    def forward(ctx, x, w, scale, bias, residual, stride, pad, dil, relu, bias_needs_grad):
        y = conv2d_forward(x, w, stride, pad, dil, scale, bias, residual, relu)
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
        g = relu_backward(dy, y) if relu else _cl(dy)
        dx = dw = db = dres = None
        # (Launching dw on a second HIP stream beside dx was measured on MI355X: 46.6 vs 45.6 ms per step —
        # slower; the contractions already hold the chip at its power-limited clock.  Kept in order.)
        if ctx.needs_input_grad[0]:
            # fold the FrozenBN scale into the weight rows once (a few MB) so the data-gradient GEMM takes
            # the direct-to-LDS path, which cannot rescale operands on the fly
            w_eff = w if scale is None else (w * scale.view(-1, 1, 1, 1)).contiguous(memory_format=CL)
            dx = conv2d_backward_data(g, w_eff, xs, stride, pad, dil)
        if ctx.needs_input_grad[1]:
            dw = conv2d_backward_weight(g, x, ws, stride, pad, dil, row_scale=scale)
        if bias_needs_grad and ctx.needs_input_grad[3]:
            db = channel_sum(g)
        if ctx.has_res and ctx.needs_input_grad[4]:
            dres = g
        return dx, dw, None, db, dres, None, None, None, None, None


def conv2d_fused(x, w, scale=None, bias=None, residual=None, stride=1, pad=0, dil=1, relu=False,
                 bias_needs_grad=False):
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
        y = _ConvFused.apply(x, w, scale, bias, residual, stride, pad, dil, relu, bias_needs_grad)
        return y[:, :o]
    return _ConvFused.apply(x, w, scale, bias, residual, stride, pad, dil, relu, bias_needs_grad)


def linear_fused(x, w, bias=None, relu=False, bias_needs_grad=True):
    """x (R, in) @ w(out, in)^T + bias, optional ReLU — the 1x1 case of conv2d_fused."""
    r, k = x.shape
    y = conv2d_fused(x.view(r, k, 1, 1), w.view(w.shape[0], k, 1, 1), None, bias, None, 1, 0, 1, relu,
                     bias_needs_grad)
    return y.view(r, w.shape[0])
