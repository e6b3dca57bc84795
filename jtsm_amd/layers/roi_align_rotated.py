"""ROIAlignRotated — Python surface of detectron2/layers/roi_align_rotated.py:10-93 on top of
libjtsm_hip.so (jtsm_roi_align_rotated_{forward,backward}_*)."""
import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from .roi_align import pooled_backward, pooled_forward


class _ROIAlignRotated(Function):
    @staticmethod
    def forward(ctx, input, roi, output_size, spatial_scale, sampling_ratio):
        ctx.save_for_backward(roi)
        ctx.output_size = _pair(output_size)
        ctx.spatial_scale = spatial_scale
        ctx.sampling_ratio = sampling_ratio
        ctx.input_shape = input.size()
        return pooled_forward("roi_align_rotated", input, roi, ctx.output_size, spatial_scale,
                              sampling_ratio, True)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        (rois,) = ctx.saved_tensors
        grad_input = pooled_backward("roi_align_rotated", grad_output, rois, ctx.output_size,
                                     ctx.spatial_scale, ctx.sampling_ratio, True, ctx.input_shape)
        return grad_input, None, None, None, None


roi_align_rotated = _ROIAlignRotated.apply


class ROIAlignRotated(nn.Module):
    def __init__(self, output_size, spatial_scale, sampling_ratio):
        """output_size (h, w); spatial_scale; sampling_ratio (0 = adaptive).  Always uses the
        continuous-coordinate (aligned) convention, like the reference."""
        super().__init__()
        self.output_size = output_size
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio

    def forward(self, input, rois):
        """
        Args:
            input: NCHW images
            rois: Bx6 boxes: (batch index, x_ctr, y_ctr, width, height, angle_degrees).
        """
        assert rois.dim() == 2 and rois.size(1) == 6
        orig_dtype = input.dtype
        if orig_dtype == torch.float16:  # the reference up-casts half (roi_align_rotated.py:79-85)
            input = input.float()
            rois = rois.float()
        return roi_align_rotated(input, rois, self.output_size, self.spatial_scale,
                                 self.sampling_ratio).to(dtype=orig_dtype)

    def __repr__(self):
        return "%s(output_size=%s, spatial_scale=%s, sampling_ratio=%s)" % (
            self.__class__.__name__, self.output_size, self.spatial_scale, self.sampling_ratio)
