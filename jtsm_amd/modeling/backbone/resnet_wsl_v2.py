"""ResNet-WS v2 ("WSR") bottom-up — the backbone the shipped JTSM configs name:
surface of projects/WSL/wsl/modeling/backbone/resnet_wsl_v2.py:122-251 (BottleneckBlock with an optional 2x2
max-pool in front), :370-429 (three-conv stem), :638-726 (build_wsl_resnet_v2_backbone).

What differs from the Base-RCNN ResNet (backbone/resnet.py):
  * the stem is three 3x3 convolutions (the first with stride 2) + MaxPool2d(2, 2);
  * no convolution is strided: the first block of res3 / res4 down-samples its INPUT with a 2x2 max-pool
    (stride 2), or — in a dilated stage — pads right/bottom by one and pools with stride 1;
  * res4 and res5 take RES5_DILATION (the DC5 configs: 2 -> stride-8 output).

MI355X mapping: identical to resnet.py — every conv+FrozenBN(+ReLU)(+shortcut) is one implicit-GEMM launch
(a whole block is one autograd node, layers/fused_blocks.py), the pools are channels-last HIP kernels
(csrc/elementwise.hip: maxpool2x2_*).  The shipped configs freeze the whole backbone (FREEZE_AT 5), so on the
training step this is a forward-only feature extractor."""
import torch.nn.functional as F

from ...layers.batch_norm import FrozenBatchNorm2d, get_norm
from ...layers.blocks import CNNBlockBase
from ...layers.elementwise import max_pool_2x2
from ...layers.fused_blocks import bottleneck_fused
from ...layers.wrappers import Conv2d
from .build import BACKBONE_REGISTRY
from .resnet import ResNet, _msra


class PooledBottleneckBlock(CNNBlockBase):
    """1x1 -> 3x3 -> 1x1 (+ projection shortcut), all stride 1; `stride` is spent in a max-pool of the input."""

    def __init__(self, in_channels, out_channels, *, bottleneck_channels, stride=1, num_groups=1, norm="BN",
                 stride_in_1x1=False, dilation=1, has_pool=False):
        super().__init__(in_channels, out_channels, stride)
        if num_groups != 1:
            raise NotImplementedError("jtsm_amd PooledBottleneckBlock: num_groups=1 only")
        self.has_pool, self.pool_stride = has_pool, stride
        if in_channels != out_channels:
            self.shortcut = Conv2d(in_channels, out_channels, kernel_size=1, stride=1, bias=False,
                                   norm=get_norm(norm, out_channels))
        else:
            self.shortcut = None
        self.conv1 = Conv2d(in_channels, bottleneck_channels, kernel_size=1, bias=False,
                            norm=get_norm(norm, bottleneck_channels), activation=F.relu)
        self.conv2 = Conv2d(bottleneck_channels, bottleneck_channels, kernel_size=3, padding=dilation, bias=False,
                            dilation=dilation, norm=get_norm(norm, bottleneck_channels), activation=F.relu)
        self.conv3 = Conv2d(bottleneck_channels, out_channels, kernel_size=1, bias=False,
                            norm=get_norm(norm, out_channels), activation=F.relu)
        for layer in (self.conv1, self.conv2, self.conv3, self.shortcut):
            if layer is not None:
                _msra(layer)

    def forward(self, x):
        if self.has_pool:
            x = max_pool_2x2(x, self.pool_stride)
        convs = [self.conv1, self.conv2, self.conv3] + ([self.shortcut] if self.shortcut is not None else [])
        if x.is_cuda and x.shape[1] % 8 == 0 and all(isinstance(c.norm, FrozenBatchNorm2d) for c in convs):
            sc = self.shortcut
            return bottleneck_fused(
                x, self.conv1.weight, self.conv1.norm.scale_bias(), self.conv2.weight, self.conv2.norm.scale_bias(),
                self.conv3.weight, self.conv3.norm.scale_bias(), sc.weight if sc is not None else None,
                sc.norm.scale_bias() if sc is not None else None, 1, 1, self.conv2.padding[0], self.conv2.dilation[0], 1)
        out = self.conv2(self.conv1(x))
        return self.conv3(out, residual=self.shortcut(x) if self.shortcut is not None else x)


class ThreeConvStem(CNNBlockBase):
    """conv3x3/2 -> conv3x3 -> conv3x3 (each + norm + ReLU) -> MaxPool2d(2, 2): stride 4."""

    def __init__(self, in_channels=3, out_channels=64, norm="BN"):
        super().__init__(in_channels, out_channels, 4)
        self.in_channels = in_channels
        self.conv1 = Conv2d(in_channels, out_channels, kernel_size=3, stride=2, padding=1, bias=False,
                            norm=get_norm(norm, out_channels), activation=F.relu)
        self.conv2 = Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=False,
                            norm=get_norm(norm, out_channels), activation=F.relu)
        self.conv3 = Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=False,
                            norm=get_norm(norm, out_channels), activation=F.relu)
        for layer in (self.conv1, self.conv2, self.conv3):
            _msra(layer)

    def forward(self, x):
        return max_pool_2x2(self.conv3(self.conv2(self.conv1(x))), 2)


@BACKBONE_REGISTRY.register()
def build_wsl_resnet_v2_backbone(cfg, input_shape):
    """cfg keys as at resnet_wsl_v2.py:638-726 (bottleneck depths; no deformable stages)."""
    norm = cfg.MODEL.RESNETS.NORM
    stem = ThreeConvStem(in_channels=input_shape.channels, out_channels=cfg.MODEL.RESNETS.STEM_OUT_CHANNELS, norm=norm)
    depth = cfg.MODEL.RESNETS.DEPTH
    if depth not in (50, 101, 152):
        raise NotImplementedError("jtsm_amd build_wsl_resnet_v2_backbone: bottleneck depths 50/101/152")
    if any(cfg.MODEL.RESNETS.DEFORM_ON_PER_STAGE):
        raise NotImplementedError("deformable conv is outside the JTSM path (SURVEY 2.1)")
    num_groups = cfg.MODEL.RESNETS.NUM_GROUPS
    bottleneck_channels = num_groups * cfg.MODEL.RESNETS.WIDTH_PER_GROUP
    in_channels = cfg.MODEL.RESNETS.STEM_OUT_CHANNELS
    out_channels = cfg.MODEL.RESNETS.RES2_OUT_CHANNELS
    res5_dilation = cfg.MODEL.RESNETS.RES5_DILATION
    assert res5_dilation in {1, 2}, "res5_dilation cannot be {}.".format(res5_dilation)
    out_features = cfg.MODEL.RESNETS.OUT_FEATURES
    blocks_per_stage = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}[depth]
    last = max({"res2": 2, "res3": 3, "res4": 4, "res5": 5}[f] for f in out_features if f != "stem")
    stages = []
    for idx, stage in enumerate(range(2, last + 1)):
        n = blocks_per_stage[idx]
        dilation = res5_dilation if stage in (4, 5) else 1
        first_stride = 2 if stage == 3 or (stage == 4 and res5_dilation == 1) else 1
        stages.append(ResNet.make_stage(
            PooledBottleneckBlock, n, in_channels=in_channels, out_channels=out_channels, norm=norm,
            bottleneck_channels=bottleneck_channels, stride_in_1x1=cfg.MODEL.RESNETS.STRIDE_IN_1X1, dilation=dilation,
            num_groups=num_groups, stride_per_block=[first_stride] + [1] * (n - 1),
            has_pool_per_block=[stage in (3, 4)] + [False] * (n - 1)))
        in_channels = out_channels
        out_channels *= 2
        bottleneck_channels *= 2
    return ResNet(stem, stages, out_features=out_features).freeze(cfg.MODEL.BACKBONE.FREEZE_AT)
