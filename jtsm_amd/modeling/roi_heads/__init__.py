from .box_head import ROI_BOX_HEAD_REGISTRY, DiscriminativeAdaptionNeck, FastRCNNConvFCHead, build_box_head
from .fast_rcnn import FastRCNNOutputLayers
from .fast_rcnn_oicr import OICROutputLayers
from .fast_rcnn_tsm import TSMOutputLayers
from .mask_head import (ROI_MASK_HEAD_REGISTRY, MaskRCNNConvUpsampleHead, MaskRCNNConvUpsampleWSLHead, build_mask_head,
                        mask_rcnn_loss)
from .roi_heads import ROI_HEADS_REGISTRY, ROIHeads, StandardROIHeads, build_roi_heads, select_foreground_proposals
from .roi_heads_jtsm import JTSMROIHeads
