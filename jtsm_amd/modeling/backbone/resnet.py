"""ResNet bottom-up — surface of detectron2/modeling/backbone/resnet.py:101-211 (BottleneckBlock),
:331-359 (BasicStem), :362-479 (ResNet), :562-648 (build_resnet_backbone).

MI355X mapping: every conv+FrozenBN(+ReLU) is one implicit-GEMM launch; the block's
`out += shortcut; relu` rides in conv3's epilogue (residual operand), so a bottleneck is 3 launches
(4 with a projection shortcut) and writes each activation exactly once.  Activations are
channels_last throughout.
"""
import torch
import torch.nn.functional as F
from torch import nn

from ...layers.batch_norm import FrozenBatchNorm2d, get_norm
from ...layers.blocks import CNNBlockBase
from ...layers.elementwise import max_pool_3x3_s2
from ...layers.fused_blocks import bottleneck_fused
from ...layers.shape_spec import ShapeSpec
from ...layers.wrappers import Conv2d
from .backbone import Backbone
from .build import BACKBONE_REGISTRY


def _msra(conv):
    """c2_msra_fill (fvcore, not in the reference tree): kaiming_normal(fan_out, relu), zero bias."""
    nn.init.kaiming_normal_(conv.weight, mode="fan_out", nonlinearity="relu")
    if conv.bias is not None:
        nn.init.constant_(conv.bias, 0)


class BottleneckBlock(CNNBlockBase):
    def __init__(self, in_channels, out_channels, *, bottleneck_channels, stride=1, num_groups=1, norm="BN",
                 stride_in_1x1=False, dilation=1):
        super().__init__(in_channels, out_channels, stride)
        if num_groups != 1:
            raise NotImplementedError("jtsm_amd BottleneckBlock: num_groups=1 only")
        if in_channels != out_channels:
            self.shortcut = Conv2d(in_channels, out_channels, kernel_size=1, stride=stride, bias=False,
                                   norm=get_norm(norm, out_channels))
        else:
            self.shortcut = None
        stride_1x1, stride_3x3 = (stride, 1) if stride_in_1x1 else (1, stride)
        self.conv1 = Conv2d(in_channels, bottleneck_channels, kernel_size=1, stride=stride_1x1, bias=False,
                            norm=get_norm(norm, bottleneck_channels), activation=F.relu)
        self.conv2 = Conv2d(bottleneck_channels, bottleneck_channels, kernel_size=3, stride=stride_3x3,
                            padding=1 * dilation, bias=False, groups=num_groups, dilation=dilation,
                            norm=get_norm(norm, bottleneck_channels), activation=F.relu)
        self.conv3 = Conv2d(bottleneck_channels, out_channels, kernel_size=1, bias=False,
                            norm=get_norm(norm, out_channels), activation=F.relu)
        for layer in [self.conv1, self.conv2, self.conv3, self.shortcut]:
            if layer is not None:
                _msra(layer)

    def _fusable(self, x):
        convs = [self.conv1, self.conv2, self.conv3] + ([self.shortcut] if self.shortcut is not None else [])
        return (x.is_cuda and x.dtype == torch.float32 and x.shape[1] % 8 == 0 and
                all(isinstance(c.norm, FrozenBatchNorm2d) and c.bias is None for c in convs))

    def forward(self, x):
        if self._fusable(x):
            # one autograd node for the block: ReLU gates and the two-path sum ride in the data-gradient epilogues
            sc = self.shortcut
            return bottleneck_fused(
                x, self.conv1.weight, self.conv1.norm.scale_bias(), self.conv2.weight, self.conv2.norm.scale_bias(),
                self.conv3.weight, self.conv3.norm.scale_bias(), sc.weight if sc is not None else None,
                sc.norm.scale_bias() if sc is not None else None, self.conv1.stride[0], self.conv2.stride[0],
                self.conv2.padding[0], self.conv2.dilation[0], sc.stride[0] if sc is not None else 1)
        out = self.conv1(x)
        out = self.conv2(out)
        shortcut = self.shortcut(x) if self.shortcut is not None else x
        return self.conv3(out, residual=shortcut)   # relu(bn(conv3(out)) + shortcut)


class BasicStem(CNNBlockBase):
    def __init__(self, in_channels=3, out_channels=64, norm="BN"):
        super().__init__(in_channels, out_channels, 4)
        self.in_channels = in_channels
        self.conv1 = Conv2d(in_channels, out_channels, kernel_size=7, stride=2, padding=3, bias=False,
                            norm=get_norm(norm, out_channels), activation=F.relu)
        _msra(self.conv1)

    def forward(self, x):
        return max_pool_3x3_s2(self.conv1(x))


class ResNet(Backbone):
    def __init__(self, stem, stages, num_classes=None, out_features=None):
        super().__init__()
        assert num_classes is None, "classification head is outside the JTSM path"
        self.stem = stem
        current_stride = self.stem.stride
        self._out_feature_strides = {"stem": current_stride}
        self._out_feature_channels = {"stem": self.stem.out_channels}
        self.stage_names, self.stages = [], []
        for i, blocks in enumerate(stages):
            assert len(blocks) > 0, len(blocks)
            name = "res" + str(i + 2)
            stage = nn.Sequential(*blocks)
            self.add_module(name, stage)
            self.stage_names.append(name)
            self.stages.append(stage)
            current_stride = int(current_stride * torch.tensor([k.stride for k in blocks]).prod().item())
            self._out_feature_strides[name] = current_stride
            self._out_feature_channels[name] = blocks[-1].out_channels
        self.stage_names = tuple(self.stage_names)
        if out_features is None:
            out_features = [name]
        self._out_features = out_features
        assert len(self._out_features)
        children = [x[0] for x in self.named_children()]
        for f in self._out_features:
            assert f in children, "Available children: {}".format(", ".join(children))

    def forward(self, x):
        assert x.dim() == 4, "ResNet takes an input of shape (N, C, H, W). Got {} instead!".format(x.shape)
        outputs = {}
        x = self.stem(x)
        if "stem" in self._out_features:
            outputs["stem"] = x
        for name, stage in zip(self.stage_names, self.stages):
            x = stage(x)
            if name in self._out_features:
                outputs[name] = x
        return outputs

    def freeze(self, freeze_at=0):
        """freeze_at=1 freezes the stem, 2 also res2, ... (resnet.py:457-479)."""
        if freeze_at >= 1:
            self.stem.freeze()
        for idx, stage in enumerate(self.stages, start=2):
            if freeze_at >= idx:
                for block in stage.children():
                    block.freeze()
        return self

    @staticmethod
    def make_stage(block_class, num_blocks, first_stride=None, *, in_channels, out_channels, **kwargs):
        if first_stride is not None:
            assert "stride" not in kwargs and "stride_per_block" not in kwargs
            kwargs["stride_per_block"] = [first_stride] + [1] * (num_blocks - 1)
        blocks = []
        for i in range(num_blocks):
            curr = {}
            for k, v in kwargs.items():
                if k.endswith("_per_block"):
                    assert len(v) == num_blocks
                    curr[k[: -len("_per_block")]] = v[i]
                else:
                    curr[k] = v
            blocks.append(block_class(in_channels=in_channels, out_channels=out_channels, **curr))
            in_channels = out_channels
        return blocks


@BACKBONE_REGISTRY.register()
def build_resnet_backbone(cfg, input_shape):
    """cfg keys as at resnet.py:562-648 (depth 50/101/152, bottleneck only)."""
    norm = cfg.MODEL.RESNETS.NORM
    stem = BasicStem(in_channels=input_shape.channels, out_channels=cfg.MODEL.RESNETS.STEM_OUT_CHANNELS, norm=norm)
    freeze_at = cfg.MODEL.BACKBONE.FREEZE_AT
    out_features = cfg.MODEL.RESNETS.OUT_FEATURES
    depth = cfg.MODEL.RESNETS.DEPTH
    num_groups = cfg.MODEL.RESNETS.NUM_GROUPS
    width_per_group = cfg.MODEL.RESNETS.WIDTH_PER_GROUP
    bottleneck_channels = num_groups * width_per_group
    in_channels = cfg.MODEL.RESNETS.STEM_OUT_CHANNELS
    out_channels = cfg.MODEL.RESNETS.RES2_OUT_CHANNELS
    stride_in_1x1 = cfg.MODEL.RESNETS.STRIDE_IN_1X1
    res5_dilation = cfg.MODEL.RESNETS.RES5_DILATION
    assert res5_dilation in {1, 2}, "res5_dilation cannot be {}.".format(res5_dilation)
    if depth not in (50, 101, 152):
        raise NotImplementedError("jtsm_amd build_resnet_backbone: bottleneck depths 50/101/152")
    if any(cfg.MODEL.RESNETS.DEFORM_ON_PER_STAGE):
        raise NotImplementedError("deformable conv is outside the JTSM path (SURVEY 2.1)")
    num_blocks_per_stage = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}[depth]
    stages = []
    out_stage_idx = [{"res2": 2, "res3": 3, "res4": 4, "res5": 5}[f] for f in out_features if f != "stem"]
    max_stage_idx = max(out_stage_idx)
    for idx, stage_idx in enumerate(range(2, max_stage_idx + 1)):
        dilation = res5_dilation if stage_idx == 5 else 1
        first_stride = 1 if idx == 0 or (stage_idx == 5 and dilation == 2) else 2
        blocks = ResNet.make_stage(
            BottleneckBlock, num_blocks_per_stage[idx], first_stride, in_channels=in_channels,
            out_channels=out_channels, norm=norm, bottleneck_channels=bottleneck_channels,
            stride_in_1x1=stride_in_1x1, dilation=dilation, num_groups=num_groups)
        in_channels = out_channels
        out_channels *= 2
        bottleneck_channels *= 2
        stages.append(blocks)
    return ResNet(stem, stages, out_features=out_features).freeze(freeze_at)
