"""ResNet-WS v2 ("WSR") bottom-up — the backbone the shipped JTSM configs name:
surface of projects/WSL/wsl/modeling/backbone/resnet_wsl_v2.py:33-119 (BasicBlock, depths 18 / 34) and :122-251
(BottleneckBlock), both with an optional 2x2 max-pool in front, :370-429 (three-conv stem), :638-726
(build_wsl_resnet_v2_backbone).

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

from ...layers.blocks import CNNBlockBase
from ...layers.elementwise import max_pool_2x2
from ...layers import fused_blocks
from ...layers.fused_blocks import bottleneck_fused
from .build import BACKBONE_REGISTRY
from .resnet import ResNet, _all_frozen, conv_norm, resnet_cfg


class _PooledBlock(CNNBlockBase):
    """A residual block none of whose convolutions is strided: `stride` is spent in a max-pool of the block INPUT
    (2x2 stride 2, or — stride 1, dilated stages — pad right/bottom by one and pool 2x2 stride 1)."""

    def __init__(self, in_channels, out_channels, stride, has_pool, pool_output=False):
        super().__init__(in_channels, out_channels, stride)
        self.has_pool, self.pool_stride = has_pool, stride
        self.pool_output = pool_output   # v1 (resnet_wsl.py:102-119,229-251) pools the block OUTPUT instead

    def pooled(self, x):
        return max_pool_2x2(x, self.pool_stride) if (self.has_pool and not self.pool_output) else x

    def pooled_out(self, y):
        return max_pool_2x2(y, self.pool_stride) if (self.has_pool and self.pool_output) else y


class PooledBasicBlock(_PooledBlock):
    """3x3 -> 3x3 (+ projection shortcut): ResNet-WS 18 / 34 (resnet_wsl_v2.py:33-119).  Two launches (three with a
    projection): the shortcut add and the final ReLU ride in conv2's epilogue."""

    def __init__(self, in_channels, out_channels, *, stride=1, norm="BN", dilation=1, has_pool=False,
                 pool_output=False):
        super().__init__(in_channels, out_channels, stride, has_pool, pool_output)
        self.shortcut = None
        if in_channels != out_channels:
            self.shortcut = conv_norm(in_channels, out_channels, 1, norm)
        self.conv1 = conv_norm(in_channels, out_channels, 3, norm, dilation=dilation, relu=True)
        self.conv2 = conv_norm(out_channels, out_channels, 3, norm, dilation=dilation, relu=True)

    def forward(self, x):
        x = self.pooled(x)
        skip = x if self.shortcut is None else self.shortcut(x)
        return self.pooled_out(self.conv2(self.conv1(x), residual=skip))


class PooledBottleneckBlock(_PooledBlock):
    """1x1 -> 3x3 -> 1x1 (+ projection shortcut), all stride 1 (resnet_wsl_v2.py:122-251)."""

    def __init__(self, in_channels, out_channels, *, bottleneck_channels, stride=1, num_groups=1, norm="BN",
                 stride_in_1x1=False, dilation=1, has_pool=False, pool_output=False):
        super().__init__(in_channels, out_channels, stride, has_pool, pool_output)
        if num_groups != 1:
            raise NotImplementedError("jtsm_amd PooledBottleneckBlock: num_groups=1 only")
        mid = bottleneck_channels
        self.shortcut = None
        if in_channels != out_channels:
            self.shortcut = conv_norm(in_channels, out_channels, 1, norm)
        self.conv1 = conv_norm(in_channels, mid, 1, norm, relu=True)
        self.conv2 = conv_norm(mid, mid, 3, norm, dilation=dilation, relu=True)
        self.conv3 = conv_norm(mid, out_channels, 1, norm, relu=True)

    def forward(self, x):
        x = self.pooled(x)
        convs = [c for c in (self.conv1, self.conv2, self.conv3, self.shortcut) if c is not None]
        if fused_blocks.ENABLED and x.is_cuda and x.shape[1] % 8 == 0 and _all_frozen(convs):
            sc = self.shortcut
            return self.pooled_out(bottleneck_fused(
                x, self.conv1.weight, self.conv1.norm.scale_bias(), self.conv2.weight, self.conv2.norm.scale_bias(),
                self.conv3.weight, self.conv3.norm.scale_bias(), sc.weight if sc is not None else None,
                sc.norm.scale_bias() if sc is not None else None, 1, 1, self.conv2.padding[0], self.conv2.dilation[0], 1))
        skip = x if self.shortcut is None else self.shortcut(x)
        return self.pooled_out(self.conv3(self.conv2(self.conv1(x)), residual=skip))


class ThreeConvStem(CNNBlockBase):
    """conv3x3/2 -> conv3x3 -> conv3x3 (each + norm + ReLU) -> MaxPool2d(2, 2): stride 4 (resnet_wsl_v2.py:370-429)."""

    def __init__(self, in_channels=3, out_channels=64, norm="BN"):
        super().__init__(in_channels, out_channels, 4)
        self.in_channels = in_channels
        self.conv1 = conv_norm(in_channels, out_channels, 3, norm, stride=2, relu=True)
        self.conv2 = conv_norm(out_channels, out_channels, 3, norm, relu=True)
        self.conv3 = conv_norm(out_channels, out_channels, 3, norm, relu=True)

    def forward(self, x):
        return max_pool_2x2(self.conv3(self.conv2(self.conv1(x))), 2)


def _build_wsl(cfg, input_shape, pooled_stages, pool_output):
    """pooled_stages: the two stages that carry a pool — (3, 4) pooling the INPUT of their first block (v2), or
    (2, 3) pooling the OUTPUT of their last block (v1).  The first of the two always pools with stride 2; the second
    with stride 2 unless RES5_DILATION is 2, in which case res4 and res5 dilate their 3x3s instead and that pool runs
    at stride 1 (the DC5 configs: stride-8 output).  No convolution of these networks is strided."""
    r, counts = resnet_cfg(cfg)
    basic = r.DEPTH in (18, 34)
    stem = ThreeConvStem(in_channels=input_shape.channels, out_channels=r.STEM_OUT_CHANNELS, norm=r.NORM)
    cin, cout, mid = r.STEM_OUT_CHANNELS, r.RES2_OUT_CHANNELS, r.NUM_GROUPS * r.WIDTH_PER_GROUP
    stages = []
    for number, n in enumerate(counts, start=2):
        dilation = r.RES5_DILATION if number in (4, 5) else 1
        pool_stride = 2 if number == pooled_stages[0] or (number == pooled_stages[1] and r.RES5_DILATION == 1) else 1
        where = n - 1 if pool_output else 0           # which block of the stage carries the stride / pool
        mark = lambda on, off: [on if i == where else off for i in range(n)]   # noqa: E731
        common = dict(in_channels=cin, out_channels=cout, norm=r.NORM, dilation=dilation, pool_output=pool_output,
                      stride_per_block=mark(pool_stride, 1), has_pool_per_block=mark(number in pooled_stages, False))
        if basic:
            stages.append(ResNet.make_stage(PooledBasicBlock, n, **common))
        else:
            stages.append(ResNet.make_stage(PooledBottleneckBlock, n, bottleneck_channels=mid,
                                            stride_in_1x1=r.STRIDE_IN_1X1, num_groups=r.NUM_GROUPS, **common))
        cin, cout, mid = cout, cout * 2, mid * 2
    return ResNet(stem, stages, out_features=r.OUT_FEATURES, pad_to_stride=not pool_output).freeze(
        cfg.MODEL.BACKBONE.FREEZE_AT)


@BACKBONE_REGISTRY.register()
def build_wsl_resnet_v2_backbone(cfg, input_shape):
    """projects/WSL/wsl/modeling/backbone/resnet_wsl_v2.py:638-726 (no deformable stages)."""
    return _build_wsl(cfg, input_shape, pooled_stages=(3, 4), pool_output=False)


@BACKBONE_REGISTRY.register()
def build_wsl_resnet_backbone(cfg, input_shape):
    """v1, projects/WSL/wsl/modeling/backbone/resnet_wsl.py:631-720 (named by jtsm_WSR_18_DC5_1x_VOC2007.yaml): the
    same blocks, but the pools sit behind the LAST block of res2 and res3."""
    return _build_wsl(cfg, input_shape, pooled_stages=(2, 3), pool_output=True)
