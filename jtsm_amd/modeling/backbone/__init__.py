from .backbone import Backbone
from .build import BACKBONE_REGISTRY, build_backbone
from .fpn import FPN, LastLevelMaxPool, build_resnet_fpn_backbone, build_wsl_resnet_v2_fpn_backbone
from .resnet import BasicStem, BottleneckBlock, ResNet, build_resnet_backbone
from .resnet_wsl_v2 import (PooledBasicBlock, PooledBottleneckBlock, ThreeConvStem, build_wsl_resnet_backbone,
                            build_wsl_resnet_v2_backbone)
