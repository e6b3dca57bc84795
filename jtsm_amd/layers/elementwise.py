"""Small bandwidth-bound kernels next to the contractions (jtsm_amd/csrc/elementwise.hip)."""
import ctypes as C

import torch

from .. import _lib as L

CL = torch.channels_last


def _dense_like(t, ref):
    """Same storage order as ref (dense)."""
    if ref.dim() == 4 and L.is_nhwc(ref):
        return t.contiguous(memory_format=CL)
    return t.contiguous()


def relu_backward(dy, y):
    L.require_gpu(dy, y)
    y = y if (y.is_contiguous() or (y.dim() == 4 and y.is_contiguous(memory_format=CL))) else y.contiguous()
    dy = _dense_like(dy, y)
    g = torch.empty_like(y)
    L.check(L.lib().jtsm_relu_backward_f32(L.ptr(dy), L.ptr(y), L.ptr(g), C.c_long(y.numel()), L.stream()),
            "relu_backward")
    return g


def channel_sum(g):
    """Sum over every axis but channels of a (N,C,H,W) channels_last or (R,C) tensor -> (C,)."""
    L.require_gpu(g)
    if g.dim() == 4:
        g = g.contiguous(memory_format=CL)
        rows, ch = g.shape[0] * g.shape[2] * g.shape[3], g.shape[1]
    else:
        g = g.contiguous()
        rows, ch = g.shape
    out = torch.empty(ch, dtype=g.dtype, device=g.device)
    L.check(L.lib().jtsm_channel_sum_f32(L.ptr(g), L.ptr(out), C.c_long(rows), ch, L.stream()), "channel_sum")
    return out
