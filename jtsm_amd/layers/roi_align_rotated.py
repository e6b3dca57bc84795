"""ROIAlignRotated — module and functional form with the signature of
detectron2/layers/roi_align_rotated.py:10-93, computed by jtsm_roi_align_rotated_{forward,backward}_*
of libjtsm_hip.so.  rois are (M, 6): (batch index, x_ctr, y_ctr, width, height, angle in degrees);
sampling always uses the continuous-coordinate ("aligned") convention."""
import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from .roi_align import pooled_backward, pooled_forward

_KIND = "roi_align_rotated"


class _ROIAlignRotated(Function):
    @staticmethod
    def forward(ctx, input, roi, output_size, spatial_scale, sampling_ratio):
        ctx.geometry = (_pair(output_size), spatial_scale, sampling_ratio, tuple(input.shape))
        ctx.save_for_backward(roi)
        return pooled_forward(_KIND, input, roi, ctx.geometry[0], spatial_scale, sampling_ratio, True)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        out_hw, scale, ratio, in_shape = ctx.geometry
        grad_input = pooled_backward(_KIND, grad_output, ctx.saved_tensors[0], out_hw, scale, ratio, True, in_shape)
        return (grad_input,) + (None,) * 4


roi_align_rotated = _ROIAlignRotated.apply


class ROIAlignRotated(nn.Module):
    def __init__(self, output_size, spatial_scale, sampling_ratio):
        """output_size: (h, w); spatial_scale: box coordinates are multiplied by it; sampling_ratio:
        samples per bin side, 0 = ceil(roi_size / output_size)."""
        super().__init__()
        self.output_size, self.spatial_scale, self.sampling_ratio = output_size, spatial_scale, sampling_ratio

    def forward(self, input, rois):
        """input: (N,C,H,W); rois: (M,6).  Half-precision inputs are computed in float32 and cast back,
        as the reference wrapper does (roi_align_rotated.py:79-85)."""
        assert rois.dim() == 2 and rois.size(1) == 6
        want = input.dtype
        if want == torch.float16:
            input, rois = input.float(), rois.float()
        out = roi_align_rotated(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio)
        return out.to(dtype=want)

    def extra_repr(self):
        return "output_size=%s, spatial_scale=%s, sampling_ratio=%s" % (self.output_size, self.spatial_scale,
                                                                        self.sampling_ratio)
