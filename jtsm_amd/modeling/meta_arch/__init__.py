from .build import META_ARCH_REGISTRY, build_model
from .mcnn import GeneralizedMCNNWSL
from .rcnn import GeneralizedRCNN, PanopticFPN
from .semantic_seg import SEM_SEG_HEADS_REGISTRY, SemSegFPNHead, build_sem_seg_head
