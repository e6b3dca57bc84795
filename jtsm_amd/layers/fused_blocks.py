"""A whole ResNet bottleneck (detectron2/modeling/backbone/resnet.py:101-211 with FrozenBN) as ONE autograd node.

Forward is the same three or four fused convolution launches as the layer-by-layer form.  The point is the
backward: inside one node the gradient chain can use the data-gradient kernel's epilogue for everything that
sits between two contractions —
    * the ReLU gates of conv1 / conv2 outputs (`relu_mask`): no separate relu_backward pass, and the gated
      gradient leaves the kernel together with its bf16 planes, ready for the next two contractions;
    * the sum of the two gradient paths into the block input (`accumulate`): no autograd add.
Per block that removes two elementwise passes over the bottleneck activations and one over the block input.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import conv as K
from .elementwise import relu_backward

CL = torch.channels_last
# False: residual blocks run layer by layer (the same kernels, one autograd node per convolution) — used by the
# frozen-gates parity test, whose forward hooks need every convolution's output
ENABLED = True


def _dgrad(g, w, scale, x_shape, stride, pad, dil, accumulate=None, relu_mask=None, emit_planes=False):
    """Data gradient through conv + FrozenBN scale (folded into the weight rows one way or the other)."""
    if K.MATH != "f32":
        return K.conv2d_backward_data(g, w, x_shape, stride, pad, dil, kscale=scale, accumulate=accumulate,
                                      relu_mask=relu_mask, emit_planes=emit_planes)
    w_eff = w if scale is None else (w * scale.view(-1, 1, 1, 1)).contiguous(memory_format=CL)
    return K.conv2d_backward_data(g, w_eff, x_shape, stride, pad, dil, accumulate=accumulate, relu_mask=relu_mask)


def _same_strides(dw, w):
    if dw is not None and dw.stride() != w.stride() and w.shape[2] == 1 and w.shape[3] == 1:
        return dw.as_strided(w.shape, w.stride())
    return dw


class _BottleneckFn(Function):
    @staticmethod
    def forward(ctx, x, w1, s1, b1, w2, s2, b2, w3, s3, b3, ws, ss, bs, stride1, stride2, pad2, dil2, stride_s):
        y1 = K.conv2d_forward(x, w1, stride1, 0, 1, s1, b1, None, True, emit_planes=True)
        y2 = K.conv2d_forward(y1, w2, stride2, pad2, dil2, s2, b2, None, True, emit_planes=True)
        sc = x if ws is None else K.conv2d_forward(x, ws, stride_s, 0, 1, ss, bs, None, False)
        y3 = K.conv2d_forward(y2, w3, 1, 0, 1, s3, b3, sc, True, emit_planes=True)
        ctx.cfg = (stride1, stride2, pad2, dil2, stride_s)
        ctx.save_for_backward(x, y1, y2, y3, w1, s1, w2, s2, w3, s3, ws, ss)
        return y3

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, y1, y2, y3, w1, s1, w2, s2, w3, s3, ws, ss = ctx.saved_tensors
        stride1, stride2, pad2, dil2, stride_s = ctx.cfg
        need = ctx.needs_input_grad
        x3 = K.MATH != "f32"
        dx = dw1 = dw2 = dw3 = dws = None
        g3 = relu_backward(dy, y3, emit_planes=x3)                       # the block's own output gate
        if need[7]:
            dw3 = K.conv2d_backward_weight(g3, y2, tuple(w3.shape), 1, 0, 1, row_scale=s3, w=w3)
        # gradient at conv2's output, gated by its ReLU in the epilogue
        d2 = _dgrad(g3, w3, s3, tuple(y2.shape), 1, 0, 1, relu_mask=y2, emit_planes=True)
        if need[4]:
            dw2 = K.conv2d_backward_weight(d2, y1, tuple(w2.shape), stride2, pad2, dil2, row_scale=s2, w=w2)
        d1 = _dgrad(d2, w2, s2, tuple(y1.shape), stride2, pad2, dil2, relu_mask=y1, emit_planes=True)
        if need[1]:
            dw1 = K.conv2d_backward_weight(d1, x, tuple(w1.shape), stride1, 0, 1, row_scale=s1, w=w1)
        if ws is not None and need[10]:
            dws = K.conv2d_backward_weight(g3, x, tuple(ws.shape), stride_s, 0, 1, row_scale=ss, w=ws)
        if need[0]:
            xs = tuple(x.shape)
            if ws is None:
                # identity shortcut: the block input receives g3 directly, added in conv1's data-gradient epilogue
                dx = _dgrad(d1, w1, s1, xs, stride1, 0, 1, accumulate=g3)
            elif stride1 == 1 and stride_s == 1:
                dxs = _dgrad(g3, ws, ss, xs, stride_s, 0, 1)
                dx = _dgrad(d1, w1, s1, xs, stride1, 0, 1, accumulate=dxs)
            else:
                # strided 1x1 pair (first block of a stage): both run as dense GEMM + scatter, summed once
                dx = _dgrad(d1, w1, s1, xs, stride1, 0, 1)
                dx = dx.add_(_dgrad(g3, ws, ss, xs, stride_s, 0, 1))
        return (dx, _same_strides(dw1, w1), None, None, _same_strides(dw2, w2), None, None, _same_strides(dw3, w3),
                None, None, _same_strides(dws, ws) if ws is not None else None, None, None, None, None, None, None, None)


def bottleneck_fused(x, w1, sb1, w2, sb2, w3, sb3, ws, sbs, stride1, stride2, pad2, dil2, stride_s):
    ss, bs = sbs if sbs is not None else (None, None)
    return _BottleneckFn.apply(x, w1, sb1[0], sb1[1], w2, sb2[0], sb2[1], w3, sb3[0], sb3[1], ws, ss, bs,
                               stride1, stride2, pad2, dil2, stride_s)
