"""ROIAlign — same Python surface as detectron2/layers/roi_align.py:14-122, computed by
libjtsm_hip.so (jtsm_amd/csrc/roi_align.hip) through the C ABI in include/jtsm_hip.h.

Layout: a channels_last (NHWC-in-memory) input takes the wavefront-per-bin NHWC kernels and
the result is returned channels_last as well; a plain contiguous NCHW input takes the
reference-layout kernels.  Either way shapes are the reference's: (N,C,H,W) -> (M,C,ph,pw).
"""
import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from .. import _lib as L

_SFX = {torch.float32: ("_f32", L.f32), torch.float64: ("_f64", L.f64), torch.float16: ("_f16", L.f32)}


def _dtype_entry(t, opname):
    try:
        return _SFX[t.dtype]
    except KeyError:
        # float32 / float64 as the reference's CUDA dispatch (ROIAlign_cuda.cu:349), plus float16 at the boundary
        # (include/jtsm_hip.h "fp16 tensors at the pooling boundary")
        raise RuntimeError('"%s" not implemented for \'%s\'' % (opname, t.dtype)) from None


def _as_layout(x):
    """Return (tensor usable by the kernels, layout flag)."""
    if L.is_nhwc(x):
        return x, L.NHWC
    return x.contiguous(), L.NCHW


def _empty_like_layout(shape, ref, layout):
    fmt = torch.channels_last if layout == L.NHWC else torch.contiguous_format
    return torch.empty(shape, dtype=ref.dtype, device=ref.device, memory_format=fmt)


def pooled_forward(kind, input, rois, out_hw, spatial_scale, sampling_ratio, aligned):
    L.require_gpu(input, rois)
    if input.dtype != rois.dtype:
        raise RuntimeError("expected input and rois to have the same dtype, got %s and %s"
                           % (input.dtype, rois.dtype))
    sfx, real = _dtype_entry(input, kind + "_forward")
    x, layout = _as_layout(input)
    rois = rois.contiguous()
    B, Cc, H, W = x.shape
    M = rois.shape[0]
    out = _empty_like_layout((M, Cc, out_hw[0], out_hw[1]), x, layout)
    if out.numel() == 0:
        return out
    fn = getattr(L.lib(), "jtsm_%s_forward%s" % (kind, sfx))
    args = [L.ptr(x), L.ptr(rois), L.ptr(out), B, Cc, H, W, M, real(spatial_scale), out_hw[0],
            out_hw[1], int(sampling_ratio)]
    if kind == "roi_align":
        args.append(int(bool(aligned)))
    if sfx == "_f16":
        import ctypes as C
        L.lib().jtsm_pool_f16_workspace_bytes.restype = C.c_size_t
        nb = L.lib().jtsm_pool_f16_workspace_bytes(C.c_long(x.numel()), C.c_long(rois.numel()), C.c_long(out.numel()),
                                                   C.c_size_t(0))
        ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
        L.check(fn(*args, layout, L.ptr(ws), C.c_size_t(nb), L.stream()), kind + "_forward")
        return out
    L.check(fn(*args, layout, L.stream()), kind + "_forward")
    return out


def pooled_backward(kind, grad, rois, out_hw, spatial_scale, sampling_ratio, aligned, in_shape):
    L.require_gpu(grad, rois)
    sfx, real = _dtype_entry(grad, kind + "_backward")
    B, Cc, H, W = in_shape
    g, layout = _as_layout(grad)
    gin = _empty_like_layout((B, Cc, H, W), g, layout)
    if gin.numel() == 0:
        return gin
    rois = rois.contiguous()
    fn = getattr(L.lib(), "jtsm_%s_backward%s" % (kind, sfx))
    args = [L.ptr(g), L.ptr(rois), L.ptr(gin), B, Cc, H, W, rois.shape[0], real(spatial_scale),
            out_hw[0], out_hw[1], int(sampling_ratio)]
    if kind == "roi_align":
        args.append(int(bool(aligned)))
    if sfx == "_f16":
        import ctypes as C
        L.lib().jtsm_pool_f16_workspace_bytes.restype = C.c_size_t
        nb = L.lib().jtsm_pool_f16_workspace_bytes(C.c_long(g.numel()), C.c_long(rois.numel()), C.c_long(gin.numel()),
                                                   C.c_size_t(0))
        ws = torch.empty(nb, dtype=torch.uint8, device=g.device)
        L.check(fn(*args, layout, L.ptr(ws), C.c_size_t(nb), L.stream()), kind + "_backward")
        return gin
    L.check(fn(*args, layout, L.stream()), kind + "_backward")
    return gin


class _ROIAlign(Function):
    @staticmethod
    def forward(ctx, input, roi, output_size, spatial_scale, sampling_ratio, aligned):
        ctx.save_for_backward(roi)
        ctx.output_size = _pair(output_size)
        ctx.spatial_scale = spatial_scale
        ctx.sampling_ratio = sampling_ratio
        ctx.input_shape = input.size()
        ctx.aligned = aligned
        return pooled_forward("roi_align", input, roi, ctx.output_size, spatial_scale,
                              sampling_ratio, aligned)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        (rois,) = ctx.saved_tensors
        grad_input = pooled_backward("roi_align", grad_output, rois, ctx.output_size,
                                     ctx.spatial_scale, ctx.sampling_ratio, ctx.aligned,
                                     ctx.input_shape)
        return grad_input, None, None, None, None, None


def roi_align(input, rois, output_size, spatial_scale=1.0, sampling_ratio=-1, aligned=False):
    return _ROIAlign.apply(input, rois, output_size, spatial_scale, sampling_ratio, aligned)


class ROIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale, sampling_ratio, aligned=True):
        """
        Args (as in the reference):
            output_size (tuple): h, w
            spatial_scale (float): scale the input boxes by this number
            sampling_ratio (int): samples per bin side; 0 = adaptive ceil(roi/bin)
            aligned (bool): True shifts the scaled box by -0.5 (pixel-centre model);
                False is the legacy Detectron behaviour.
        """
        super().__init__()
        self.output_size = output_size
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio
        self.aligned = aligned

    def forward(self, input, rois):
        """
        Args:
            input: NCHW images
            rois: Bx5 boxes. First column is the index into N. The other 4 columns are xyxy.
        """
        assert rois.dim() == 2 and rois.size(1) == 5
        return roi_align(input, rois.to(dtype=input.dtype), self.output_size, self.spatial_scale,
                         self.sampling_ratio, self.aligned)

    def __repr__(self):
        return "%s(output_size=%s, spatial_scale=%s, sampling_ratio=%s, aligned=%s)" % (
            self.__class__.__name__, self.output_size, self.spatial_scale, self.sampling_ratio,
            self.aligned)
