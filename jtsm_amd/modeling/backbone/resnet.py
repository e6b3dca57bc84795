"""ResNet bottom-up behind the reference's names — API surface of detectron2/modeling/backbone/resnet.py
(:26-98 BasicBlock, :101-211 BottleneckBlock, :331-359 BasicStem, :362-479 ResNet, :562-648 build_resnet_backbone):
same class names, constructor keywords, module names (stem.conv1, resN.K.convM / shortcut) and therefore the same
state_dict keys; everything else is this repo's own construction.

MI355X mapping: every conv + FrozenBN (+ ReLU) (+ shortcut add) is ONE implicit-GEMM launch (layers/wrappers.py);
a whole residual block is one autograd node whenever its norms are frozen (layers/fused_blocks.py), so the ReLU
gates and the two-path gradient sum ride in data-gradient epilogues.  Activations are channels_last throughout.
A network is described by a small table (blocks per stage, first stride, dilation) and built by one loop.
"""
import torch
import torch.nn.functional as F
from torch import nn

from ...layers.grad_fan import fan_out
from ...layers.batch_norm import FrozenBatchNorm2d, get_norm
from ...layers.blocks import CNNBlockBase
from ...layers.elementwise import max_pool_3x3_s2
from ...layers import fused_blocks
from ...layers.fused_blocks import bottleneck_fused
from ...layers.wrappers import Conv2d
from .backbone import Backbone
from .build import BACKBONE_REGISTRY

# blocks in res2..res5 per depth; depths below 50 use the two-conv BasicBlock
BLOCKS_PER_STAGE = {18: (2, 2, 2, 2), 34: (3, 4, 6, 3), 50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}
_STAGE_NUMBER = {"res2": 2, "res3": 3, "res4": 4, "res5": 5}


def _msra(conv):
    """c2_msra_fill (fvcore, not in the reference tree): kaiming_normal(fan_out, relu), zero bias."""
    nn.init.kaiming_normal_(conv.weight, mode="fan_out", nonlinearity="relu")
    if conv.bias is not None:
        nn.init.constant_(conv.bias, 0)


def conv_norm(cin, cout, k, norm, *, stride=1, dilation=1, relu=False):
    """Bias-free k x k convolution ('same' padding for its dilation) + norm (+ ReLU), msra-initialised."""
    conv = Conv2d(cin, cout, kernel_size=k, stride=stride, padding=(k // 2) * dilation, dilation=dilation, bias=False,
                  norm=get_norm(norm, cout), activation=F.relu if relu else None)
    _msra(conv)
    return conv


def _all_frozen(convs):
    return all(isinstance(c.norm, FrozenBatchNorm2d) and c.bias is None for c in convs)


class BasicBlock(CNNBlockBase):
    """Two 3x3 convolutions + (projection) shortcut — ResNet-18/34 (resnet.py:26-98)."""

    def __init__(self, in_channels, out_channels, *, stride=1, norm="BN"):
        super().__init__(in_channels, out_channels, stride)
        self.shortcut = None
        if in_channels != out_channels:
            self.shortcut = conv_norm(in_channels, out_channels, 1, norm, stride=stride)
        self.conv1 = conv_norm(in_channels, out_channels, 3, norm, stride=stride, relu=True)
        self.conv2 = conv_norm(out_channels, out_channels, 3, norm, relu=True)   # ReLU after the shortcut add

    def forward(self, x):
        skip = x if self.shortcut is None else self.shortcut(x)
        return self.conv2(self.conv1(x), residual=skip)


class BottleneckBlock(CNNBlockBase):
    """1x1 -> 3x3 -> 1x1 + (projection) shortcut — ResNet-50/101/152 (resnet.py:101-211)."""

    def __init__(self, in_channels, out_channels, *, bottleneck_channels, stride=1, num_groups=1, norm="BN",
                 stride_in_1x1=False, dilation=1):
        super().__init__(in_channels, out_channels, stride)
        if num_groups != 1:
            raise NotImplementedError("jtsm_amd BottleneckBlock: num_groups=1 only")
        s1, s3 = (stride, 1) if stride_in_1x1 else (1, stride)
        mid = bottleneck_channels
        self.shortcut = None
        if in_channels != out_channels:
            self.shortcut = conv_norm(in_channels, out_channels, 1, norm, stride=stride)
        self.conv1 = conv_norm(in_channels, mid, 1, norm, stride=s1, relu=True)
        self.conv2 = conv_norm(mid, mid, 3, norm, stride=s3, dilation=dilation, relu=True)
        self.conv3 = conv_norm(mid, out_channels, 1, norm, relu=True)           # ReLU after the shortcut add

    def _members(self):
        return [c for c in (self.conv1, self.conv2, self.conv3, self.shortcut) if c is not None]

    def forward(self, x):
        if fused_blocks.ENABLED and x.is_cuda and x.dtype == torch.float32 and x.shape[1] % 8 == 0 and \
                _all_frozen(self._members()):
            # one autograd node for the block: ReLU gates and the two-path sum ride in the data-gradient epilogues
            sc = self.shortcut
            return bottleneck_fused(
                x, self.conv1.weight, self.conv1.norm.scale_bias(), self.conv2.weight, self.conv2.norm.scale_bias(),
                self.conv3.weight, self.conv3.norm.scale_bias(), sc.weight if sc is not None else None,
                sc.norm.scale_bias() if sc is not None else None, self.conv1.stride[0], self.conv2.stride[0],
                self.conv2.padding[0], self.conv2.dilation[0], sc.stride[0] if sc is not None else 1)
        skip = x if self.shortcut is None else self.shortcut(x)
        return self.conv3(self.conv2(self.conv1(x)), residual=skip)   # relu(bn(conv3(.)) + skip)


class BasicStem(CNNBlockBase):
    """7x7 stride-2 convolution + norm + ReLU, then max-pool 3x3 stride 2: stride 4 (resnet.py:331-359)."""

    def __init__(self, in_channels=3, out_channels=64, norm="BN"):
        super().__init__(in_channels, out_channels, 4)
        self.in_channels = in_channels
        self.conv1 = conv_norm(in_channels, out_channels, 7, norm, stride=2, relu=True)

    def forward(self, x):
        return max_pool_3x3_s2(self.conv1(x))


class ResNet(Backbone):
    """stem, then stages `res2`, `res3`, ...; returns the requested subset of {"stem", "res2", ...}."""

    def __init__(self, stem, stages, num_classes=None, out_features=None, pad_to_stride=False):
        """pad_to_stride: report the last stage's stride as `size_divisibility` (the WSL v2 network does,
        projects/WSL/wsl/modeling/backbone/resnet_wsl_v2.py:474,495-497; detectron2's own ResNet reports 0)."""
        super().__init__()
        if num_classes is not None:
            raise NotImplementedError("the classification head is outside the JTSM path")
        self.stem = stem
        channels, strides = {"stem": stem.out_channels}, {"stem": stem.stride}
        stride, order = stem.stride, []
        for number, blocks in enumerate(stages, start=2):
            blocks = list(blocks)
            if not blocks:
                raise ValueError("ResNet: stage res%d has no blocks" % number)
            name = "res%d" % number
            self.add_module(name, nn.Sequential(*blocks))
            for b in blocks:
                stride *= b.stride
            channels[name], strides[name] = blocks[-1].out_channels, stride
            order.append(name)
        self.stage_names = tuple(order)
        wanted = list(out_features) if out_features else [order[-1]]
        unknown = [f for f in wanted if f not in channels]
        if unknown:
            raise ValueError("ResNet: unknown out_features %s (have %s)" % (unknown, sorted(channels)))
        self._declare_outputs(wanted, {f: channels[f] for f in wanted}, {f: strides[f] for f in wanted})
        self._size_divisibility = stride if pad_to_stride else 0

    @property
    def size_divisibility(self):
        return self._size_divisibility

    @property
    def stages(self):
        return [getattr(self, n) for n in self.stage_names]

    def forward(self, x):
        if x.dim() != 4:
            raise ValueError("ResNet takes an input of shape (N, C, H, W), got %s" % (tuple(x.shape),))
        keep = set(self._out_features)
        out = {}
        x = self.stem(x)
        if "stem" in keep:
            out["stem"] = x
        for name in self.stage_names:
            x = self._run_stage(getattr(self, name), x)
            if name in keep:
                if name != self.stage_names[-1]:
                    # read twice — by whoever takes the feature (an FPN lateral) and by the next stage: one view each, so
                    # that the two gradient terms meet in one map (layers/grad_fan.py)
                    out[name], x = fan_out(x, 2)
                else:
                    out[name] = x
        return out

    @staticmethod
    def _run_stage(stage, x):
        """The stage's blocks in order.  In the fp16 arithmetic the identity-shortcut blocks behind the first one run
        as ONE node whose activations are fp16 planes only (layers/fused_blocks.py: _IdentityChain16Fn)."""
        blocks = list(stage.children())
        rest = blocks[1:]
        if rest and all(isinstance(b, BottleneckBlock) and _all_frozen(b._members()) for b in blocks) and \
                fused_blocks.identity_chain_ok(x, rest):
            y = blocks[0](x)
            if fused_blocks.identity_chain_ok(y, rest):
                return fused_blocks.identity_chain_fused(y, rest)
            for b in rest:
                y = b(y)
            return y
        return stage(x)

    def freeze(self, freeze_at=0):
        """freeze_at = 1 freezes the stem, 2 also res2, ... (resnet.py:457-479)."""
        if freeze_at >= 1:
            self.stem.freeze()
        for name in self.stage_names:
            if freeze_at >= _STAGE_NUMBER[name]:
                for block in getattr(self, name).children():
                    block.freeze()
        return self

    @staticmethod
    def make_stage(block_class, num_blocks, first_stride=None, *, in_channels, out_channels, **kwargs):
        """`num_blocks` blocks of `block_class`; keyword `foo_per_block=[...]` gives block i its own `foo`."""
        if first_stride is not None:
            if "stride" in kwargs or "stride_per_block" in kwargs:
                raise ValueError("make_stage: give first_stride or stride(_per_block), not both")
            kwargs["stride_per_block"] = [first_stride] + [1] * (num_blocks - 1)
        suffix = "_per_block"
        varying = {k[:-len(suffix)]: v for k, v in kwargs.items() if k.endswith(suffix)}
        shared = {k: v for k, v in kwargs.items() if not k.endswith(suffix)}
        for k, v in varying.items():
            if len(v) != num_blocks:
                raise ValueError("make_stage: %s%s has %d entries for %d blocks" % (k, suffix, len(v), num_blocks))
        blocks, cin = [], in_channels
        for i in range(num_blocks):
            blocks.append(block_class(in_channels=cin, out_channels=out_channels, **shared,
                                      **{k: v[i] for k, v in varying.items()}))
            cin = out_channels
        return blocks


def resnet_cfg(cfg):
    """The MODEL.RESNETS keys every ResNet builder reads, validated once."""
    r = cfg.MODEL.RESNETS
    if r.DEPTH not in BLOCKS_PER_STAGE:
        raise ValueError("MODEL.RESNETS.DEPTH must be one of %s, got %r" % (sorted(BLOCKS_PER_STAGE), r.DEPTH))
    if any(r.DEFORM_ON_PER_STAGE):
        raise NotImplementedError("deformable conv is outside the JTSM path (SURVEY 2.1)")
    assert r.RES5_DILATION in {1, 2}, "res5_dilation cannot be {}.".format(r.RES5_DILATION)
    if r.DEPTH in (18, 34):
        assert r.RES2_OUT_CHANNELS == 64, "Must set MODEL.RESNETS.RES2_OUT_CHANNELS = 64 for R18/R34"
        assert r.NUM_GROUPS == 1, "Must set MODEL.RESNETS.NUM_GROUPS = 1 for R18/R34"
    last = max(_STAGE_NUMBER[f] for f in r.OUT_FEATURES if f != "stem")
    return r, BLOCKS_PER_STAGE[r.DEPTH][:last - 1]


@BACKBONE_REGISTRY.register()
def build_resnet_backbone(cfg, input_shape):
    """cfg keys as at resnet.py:562-648.  Stage table: res2 keeps the stem's stride, every later stage halves the
    map in its first block — except a dilated res5 (DC5), which keeps res4's stride and dilates its 3x3s by 2."""
    r, counts = resnet_cfg(cfg)
    basic = r.DEPTH in (18, 34)
    if basic:
        assert r.RES5_DILATION == 1, "Must set MODEL.RESNETS.RES5_DILATION = 1 for R18/R34"
    stem = BasicStem(in_channels=input_shape.channels, out_channels=r.STEM_OUT_CHANNELS, norm=r.NORM)
    cin, cout, mid = r.STEM_OUT_CHANNELS, r.RES2_OUT_CHANNELS, r.NUM_GROUPS * r.WIDTH_PER_GROUP
    stages = []
    for number, n in enumerate(counts, start=2):
        dilated = number == 5 and r.RES5_DILATION == 2
        first_stride = 1 if (number == 2 or dilated) else 2
        if basic:
            stage = ResNet.make_stage(BasicBlock, n, first_stride, in_channels=cin, out_channels=cout, norm=r.NORM)
        else:
            stage = ResNet.make_stage(BottleneckBlock, n, first_stride, in_channels=cin, out_channels=cout,
                                      norm=r.NORM, bottleneck_channels=mid, stride_in_1x1=r.STRIDE_IN_1X1,
                                      dilation=2 if dilated else 1, num_groups=r.NUM_GROUPS)
        stages.append(stage)
        cin, cout, mid = cout, cout * 2, mid * 2
    return ResNet(stem, stages, out_features=r.OUT_FEATURES).freeze(cfg.MODEL.BACKBONE.FREEZE_AT)
