"""MOIPool — Python surface of projects/WSL/wsl/layers/moi_pool.py:10-88 on top of
libjtsm_hip.so (jtsm_moi_pool_{forward,backward}_f32)."""
import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from .. import _lib as L
from .roi_align import _as_layout, _empty_like_layout


def moi_pool_forward(input, rois, spatial_scale, pooled_h, pooled_w, oh_labels, superpixels):
    L.require_gpu(input, rois, oh_labels, superpixels)
    if input.dtype not in (torch.float32, torch.float16):
        raise RuntimeError('"MOIPool_forward" is implemented for float32 and float16, got %s' % input.dtype)
    if input.dtype != rois.dtype:
        raise RuntimeError("expected input and rois to have the same dtype")
    x, layout = _as_layout(input)
    rois = rois.contiguous()
    oh = oh_labels.to(torch.int32).contiguous()
    sp = superpixels.to(torch.int32).contiguous()
    B, Cc, H, W = x.shape
    M, Lw = rois.shape[0], oh.shape[1]
    out = _empty_like_layout((M, Cc, pooled_h, pooled_w), x, layout)
    fmt = torch.channels_last if layout == L.NHWC else torch.contiguous_format
    arg = torch.empty((M, Cc, pooled_h, pooled_w), dtype=torch.int32, device=x.device,
                      memory_format=fmt)
    if out.numel() == 0:
        return out, arg
    if input.dtype == torch.float16:   # fp16 at the boundary (MOIPool_cuda.cu:400 dispatches on half too)
        import ctypes as C
        lib = L.lib()
        lib.jtsm_moi_pool_f16_workspace_bytes.restype = C.c_size_t
        nb = lib.jtsm_moi_pool_f16_workspace_bytes(B, Cc, H, W, M, Lw, pooled_h, pooled_w)
        ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
        L.check(lib.jtsm_moi_pool_forward_f16(
            L.ptr(x), L.ptr(rois), L.ptr(oh), L.ptr(sp), L.ptr(out), L.ptr(arg), L.ptr(ws), C.c_size_t(nb), B, Cc, H, W,
            M, Lw, sp.shape[1], sp.shape[2], L.f32(spatial_scale), pooled_h, pooled_w, layout, L.stream()),
            "moi_pool_forward_f16")
        return out, arg
    nbytes = L.lib().jtsm_moi_pool_workspace_bytes(B, H, W, M, Lw)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    L.check(L.lib().jtsm_moi_pool_forward_f32(
        L.ptr(x), L.ptr(rois), L.ptr(oh), L.ptr(sp), L.ptr(out), L.ptr(arg), L.ptr(ws), B, Cc, H,
        W, M, Lw, sp.shape[1], sp.shape[2], L.f32(spatial_scale), pooled_h, pooled_w, layout,
        L.stream()), "moi_pool_forward")
    return out, arg


def moi_pool_backward(grad, rois, argmax, spatial_scale, pooled_h, pooled_w, B, Cc, H, W):
    L.require_gpu(grad, rois, argmax)
    # grad and argmax must share one storage order; argmax decides (it was produced by forward)
    layout = L.NHWC if L.is_nhwc(argmax) else L.NCHW
    fmt = torch.channels_last if layout == L.NHWC else torch.contiguous_format
    g = grad.contiguous(memory_format=fmt)
    a = argmax.contiguous(memory_format=fmt)
    gin = _empty_like_layout((B, Cc, H, W), g, layout)
    if gin.numel() == 0:
        return gin
    if g.dtype == torch.float16:
        import ctypes as C
        lib = L.lib()
        lib.jtsm_pool_f16_workspace_bytes.restype = C.c_size_t
        nb = lib.jtsm_pool_f16_workspace_bytes(C.c_long(g.numel()), C.c_long(rois.numel()), C.c_long(gin.numel()),
                                               C.c_size_t(0))
        ws = torch.empty(nb, dtype=torch.uint8, device=g.device)
        L.check(lib.jtsm_moi_pool_backward_f16(
            L.ptr(g), L.ptr(rois.contiguous()), L.ptr(a), L.ptr(gin), L.ptr(ws), C.c_size_t(nb), B, Cc, H, W,
            rois.shape[0], pooled_h, pooled_w, layout, L.stream()), "moi_pool_backward_f16")
        return gin
    L.check(L.lib().jtsm_moi_pool_backward_f32(
        L.ptr(g), L.ptr(rois.contiguous()), L.ptr(a), L.ptr(gin), B, Cc, H, W, rois.shape[0],
        pooled_h, pooled_w, layout, L.stream()), "moi_pool_backward")
    return gin


class _MOIPool(Function):
    @staticmethod
    def forward(ctx, input, roi, output_size, spatial_scale, oh_labels, superpixels):
        ctx.output_size = _pair(output_size)
        ctx.spatial_scale = spatial_scale
        ctx.input_shape = input.size()
        output, argmax = moi_pool_forward(input, roi, spatial_scale, ctx.output_size[0],
                                          ctx.output_size[1], oh_labels, superpixels)
        ctx.save_for_backward(roi, argmax)
        ctx.mark_non_differentiable(argmax)
        ctx.set_materialize_grads(False)   # no zero-filled "gradient" for the arg-max output
        return output, argmax

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output, _grad_argmax=None):
        if grad_output is None:
            return None, None, None, None, None, None
        rois, argmax = ctx.saved_tensors
        bs, ch, h, w = ctx.input_shape
        grad_input = moi_pool_backward(grad_output, rois, argmax, ctx.spatial_scale,
                                       ctx.output_size[0], ctx.output_size[1], bs, ch, h, w)
        return grad_input, None, None, None, None, None


moi_pool = _MOIPool.apply


class MOIPool(nn.Module):
    def __init__(self, output_size, spatial_scale):
        """output_size (h, w); spatial_scale: multiply boxes by this before rounding."""
        super().__init__()
        self.output_size = output_size
        self.spatial_scale = spatial_scale

    def forward(self, input, rois, oh_labels, superpixels):
        """
        Args:
            input: NCHW features
            rois: Bx5 boxes (batch index, x0, y0, x1, y1)
            oh_labels: (B, L) int32, 1 where superpixel id belongs to the box
            superpixels: (N, Hs, Ws) int32 superpixel id per image pixel
        Returns: (output, argmax)
        """
        assert rois.dim() == 2 and rois.size(1) == 5
        return moi_pool(input, rois, self.output_size, self.spatial_scale, oh_labels, superpixels)

    def __repr__(self):
        return "%s(output_size=%s, spatial_scale=%s)" % (
            self.__class__.__name__, self.output_size, self.spatial_scale)
