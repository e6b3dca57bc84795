from .backbone import Backbone
from .build import BACKBONE_REGISTRY, build_backbone
from .fpn import FPN, LastLevelMaxPool, build_resnet_fpn_backbone
from .resnet import BasicStem, BottleneckBlock, ResNet, build_resnet_backbone
