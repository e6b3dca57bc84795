"""Conv2d / Linear / ConvTranspose2d wrappers — surface of detectron2/layers/wrappers.py:16-141.

``Conv2d(*args, norm=None, activation=None)`` keeps the reference's constructor and parameter
names (weight, bias, norm.*), but its forward is ONE launch of the fp32-MFMA implicit GEMM with the
norm (FrozenBatchNorm2d), bias and ReLU folded into the epilogue.  Norms/activations the epilogue
cannot express (GroupNorm, non-ReLU) run after it as separate modules, like in the reference.
"""
import torch
import torch.nn.functional as F
from torch import nn

from .batch_norm import FrozenBatchNorm2d
from .conv import conv2d_fused, conv_transpose2x2_fused, conv_transpose2x2_ok, linear_fused

CL = torch.channels_last


def cat(tensors, dim=0):
    """torch.cat that skips the copy for a single-element list (wrappers.py:16-23)."""
    assert isinstance(tensors, (list, tuple))
    if len(tensors) == 1:
        return tensors[0]
    return torch.cat(tensors, dim)


def nonzero_tuple(x):
    if x.dim() == 0:
        return x.unsqueeze(0).nonzero().unbind(1)
    return x.nonzero().unbind(1)


def _is_relu(act):
    return act is F.relu or act is F.relu_ or isinstance(act, nn.ReLU)


class Conv2d(nn.Conv2d):
    def __init__(self, *args, **kwargs):
        norm = kwargs.pop("norm", None)
        activation = kwargs.pop("activation", None)
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation
        if self.groups != 1 or self.padding_mode != "zeros":
            raise NotImplementedError("jtsm_amd Conv2d: groups=1, zero padding only (all the JTSM path uses)")
        for a in (self.stride, self.padding, self.dilation):
            if a[0] != a[1]:
                raise NotImplementedError("jtsm_amd Conv2d: square stride/padding/dilation only")
        # OHWI storage: what the kernels read; logical shape stays (O,I,kh,kw)
        self.weight.data = self.weight.data.contiguous(memory_format=CL)

    def forward(self, x, residual=None):
        """residual (same shape as the output) is added before the activation — the bottleneck's
        `out += shortcut; relu` (resnet.py:203-210) in the same launch."""
        fuse_norm = isinstance(self.norm, FrozenBatchNorm2d)
        relu = _is_relu(self.activation) and (self.norm is None or fuse_norm)
        scale = bias = None
        bias_grad = False
        if fuse_norm:
            scale, bias = self.norm.scale_bias()
            if self.bias is not None:
                bias = bias + self.bias * scale
        elif self.bias is not None:
            bias, bias_grad = self.bias, True
        if x.shape[1] % 4:  # RGB stem: zero-pad channels to a multiple of 4 (kernel requirement)
            padc = 4 - x.shape[1] % 4
            x = F.pad(x, (0, 0, 0, 0, 0, padc))
            w = F.pad(self.weight, (0, 0, 0, 0, 0, padc))
        else:
            w = self.weight
        y = conv2d_fused(x, w, scale, bias, residual, self.stride[0], self.padding[0], self.dilation[0], relu,
                         bias_grad, emit_dx_planes=getattr(self, "emit_dx_planes", False))
        if isinstance(self.norm, nn.GroupNorm) and self.norm.affine and y.is_cuda and y.dtype == torch.float32:
            # GroupNorm (+ the following ReLU) as one channels-last pass (jtsm_amd/csrc/semseg_ops.hip)
            from .elementwise import group_norm_relu
            act_relu = _is_relu(self.activation)
            y = group_norm_relu(y, self.norm.weight, self.norm.bias, self.norm.num_groups, self.norm.eps, act_relu)
            if self.activation is not None and not act_relu:
                y = self.activation(y)
            return y
        if self.norm is not None and not fuse_norm:
            y = self.norm(y)
        if self.activation is not None and not relu:
            y = self.activation(y)
        return y


class Linear(nn.Linear):
    """nn.Linear on the MFMA GEMM; `relu=True` folds the following ReLU into the epilogue."""

    def forward(self, x, relu=False):
        return linear_fused(x, self.weight, self.bias, relu, self.bias is not None)


class ConvTranspose2d(nn.ConvTranspose2d):
    """kernel 2, stride 2, padding 0 only (the mask head's upsampler, mask_head.py:303-305): each of
    the 4 output phases is a 1x1 convolution, so the layer is ONE GEMM with 4*out columns whose epilogue
    does the pixel shuffle (exact-fp32 arithmetic: the GEMM, then a shuffle copy)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        # (in, kh, kw, out) storage: what the kernels read; the logical shape stays (in, out, kh, kw)
        self.weight.data = self.weight.data.contiguous(memory_format=CL)

    def forward(self, x, relu=False):
        if self.kernel_size != (2, 2) or self.stride != (2, 2) or self.padding != (0, 0):
            raise NotImplementedError("jtsm_amd ConvTranspose2d: kernel=stride=2, padding=0 only")
        if x.is_cuda and conv_transpose2x2_ok(x, self.weight):
            # the GEMM's epilogue writes the four output pixels of every input pixel itself; the backward is the
            # forward / weight-gradient role of the 2x2 stride-2 convolution this layer transposes (layers/conv.py)
            return conv_transpose2x2_fused(x, self.weight, self.bias, relu)
        n, c, h, w = x.shape
        o = self.out_channels
        wl = self.weight.permute(2, 3, 1, 0).reshape(4 * o, c).contiguous()   # [(dy,dx,o)][i] (a copy: the parameter
        # is stored (in, kh, kw, out))
        b = self.bias.repeat(4) if self.bias is not None else None
        rows = x.permute(0, 2, 3, 1).reshape(n * h * w, c)                # NHWC rows (view when channels_last)
        z = linear_fused(rows, wl, b, relu, b is not None)                # (n*h*w, 4*o)
        z = z.view(n, h, w, 2, 2, o).permute(0, 1, 3, 2, 4, 5).reshape(n, 2 * h, 2 * w, o)
        return z.permute(0, 3, 1, 2)                                      # logical NCHW, NHWC memory
