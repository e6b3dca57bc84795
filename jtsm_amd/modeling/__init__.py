"""Model components of the JTSM hot path behind the reference's registries
(detectron2/modeling/__init__.py; projects/WSL/wsl/modeling)."""
from .backbone import BACKBONE_REGISTRY, FPN, Backbone, ResNet, build_backbone, build_resnet_fpn_backbone
from .anchor_generator import ANCHOR_GENERATOR_REGISTRY, DefaultAnchorGenerator, build_anchor_generator
from .meta_arch import (META_ARCH_REGISTRY, SEM_SEG_HEADS_REGISTRY, GeneralizedMCNNWSL, GeneralizedRCNN, PanopticFPN,
                        SemSegFPNHead, build_model, build_sem_seg_head)
from .proposal_generator import PROPOSAL_GENERATOR_REGISTRY, RPN, RPN_HEAD_REGISTRY, build_proposal_generator
from .poolers import ROIPooler
from .roi_heads import (ROI_BOX_HEAD_REGISTRY, ROI_HEADS_REGISTRY, ROI_MASK_HEAD_REGISTRY, JTSMROIHeads,
                        StandardROIHeads, build_roi_heads)
