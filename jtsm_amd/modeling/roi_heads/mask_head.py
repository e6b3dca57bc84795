"""Mask head — surface of projects/WSL/wsl/modeling/roi_heads/mask_head.py:23-103 (mask_rcnn_loss) and
:266-343 (MaskRCNNConvUpsampleWSLHead, whose `layers` returns logits AND the upsampled features).

4 x [conv3x3 + ReLU] and the 1x1 predictor are MFMA launches with bias/ReLU in the epilogue; the
2x2/stride-2 ConvTranspose is one GEMM whose epilogue does the pixel shuffle; in the default arithmetic the whole
tower is ONE autograd node whose backward keeps the ReLU gates in the data-gradient epilogues
(jtsm_amd/layers/fused_blocks.py: _MaskTowerFn)."""
import torch
import torch.nn.functional as F
from torch import nn

from ...layers.shape_spec import ShapeSpec
from ...layers.fused_blocks import mask_tower_fused, mask_tower_ok
from ...layers.wrappers import Conv2d, ConvTranspose2d, cat
from ...layers.postprocess import mask_probs
from ...layers.wsl_losses import mask_bce_loss
from ...utils.registry import Registry

ROI_MASK_HEAD_REGISTRY = Registry("ROI_MASK_HEAD")


def build_mask_head(cfg, input_shape):
    return ROI_MASK_HEAD_REGISTRY.get(cfg.MODEL.ROI_MASK_HEAD.NAME)(cfg, input_shape)


def mask_rcnn_loss(pred_mask_logits, gt_classes, gt_masks_bool):
    """BCE-with-logits (mean) between the gt-class channel of (N,C,M,M) logits and (N,M,M) bool targets.
    (The reference takes Instances and rasterises targets inside; here targets come pre-cropped.)"""
    total_num_masks = pred_mask_logits.size(0)
    if total_num_masks == 0:
        return pred_mask_logits.sum() * 0
    return mask_bce_loss(pred_mask_logits, gt_classes, gt_masks_bool)     # one HIP launch each way (no CPU path)


@torch.no_grad()
def mask_rcnn_inference(pred_mask_logits, pred_instances):
    """mask_head.py:106-147: sigmoid of each detection's predicted-class channel, stored per image as
    `pred_masks` (Ri, 1, M, M).  `pred_mask_logits` may be a list of heads' logits: they are averaged first
    (roi_heads_jtsm.py:949-961) inside the same launch."""
    heads = list(pred_mask_logits) if isinstance(pred_mask_logits, (list, tuple)) else [pred_mask_logits]
    if heads[0].size(0) == 0:
        probs = heads[0].new_zeros((0, 1) + tuple(heads[0].shape[2:]))
    else:
        probs = mask_probs(heads, cat([i.pred_classes for i in pred_instances]))
    for prob, instances in zip(probs.split([len(i) for i in pred_instances], dim=0), pred_instances):
        instances.pred_masks = prob


@ROI_MASK_HEAD_REGISTRY.register()
class MaskRCNNConvUpsampleWSLHead(nn.Module):
    def __init__(self, cfg_or_shape, input_shape: ShapeSpec = None, *, num_classes=None, conv_dims=None,
                 conv_norm=""):
        super().__init__()
        if input_shape is not None:
            cfg = cfg_or_shape
            conv_dims = [cfg.MODEL.ROI_MASK_HEAD.CONV_DIM] * (cfg.MODEL.ROI_MASK_HEAD.NUM_CONV + 1)
            conv_norm = cfg.MODEL.ROI_MASK_HEAD.NORM
            num_classes = 1 if cfg.MODEL.ROI_MASK_HEAD.CLS_AGNOSTIC_MASK else cfg.MODEL.ROI_HEADS.NUM_CLASSES
        else:
            input_shape = cfg_or_shape
        assert len(conv_dims) >= 1, "conv_dims have to be non-empty!"
        if conv_norm:
            raise NotImplementedError("mask head norm '%s' is not used on the JTSM path" % conv_norm)
        self.conv_norm_relus = []
        cur_channels = input_shape.channels
        for k, conv_dim in enumerate(conv_dims[:-1]):
            conv = Conv2d(cur_channels, conv_dim, kernel_size=3, stride=1, padding=1, bias=True, activation=F.relu)
            self.add_module("mask_fcn{}".format(k + 1), conv)
            self.conv_norm_relus.append(conv)
            cur_channels = conv_dim
        self.deconv = ConvTranspose2d(cur_channels, conv_dims[-1], kernel_size=2, stride=2, padding=0)
        cur_channels = conv_dims[-1]
        self.predictor = Conv2d(cur_channels, num_classes, kernel_size=1, stride=1, padding=0)
        for layer in self.conv_norm_relus + [self.deconv]:
            nn.init.kaiming_normal_(layer.weight, mode="fan_out", nonlinearity="relu")
            nn.init.constant_(layer.bias, 0)
        nn.init.normal_(self.predictor.weight, std=0.001)
        nn.init.constant_(self.predictor.bias, 0)

    def layers(self, x):
        if mask_tower_ok(x, self.conv_norm_relus, self.deconv, self.predictor):
            # one autograd node; `return_features = False` (set by a caller that uses the logits only, as
            # JTSMROIHeads does) spares the fp32 copy of the upsampled features: the second value is then None
            return mask_tower_fused(x, self.conv_norm_relus, self.deconv, self.predictor,
                                    getattr(self, "return_features", True))
        for layer in self.conv_norm_relus:
            x = layer(x)
        x = self.deconv(x, relu=True)
        return self.predictor(x), x


@ROI_MASK_HEAD_REGISTRY.register()
class MaskRCNNConvUpsampleHead(MaskRCNNConvUpsampleWSLHead):
    """detectron2/modeling/roi_heads/mask_head.py:201-290 — the same layers (state_dict keys included) as the WSL
    variant, which only adds the second return value of `layers`."""

    def forward(self, x):
        return self.layers(x)[0]
