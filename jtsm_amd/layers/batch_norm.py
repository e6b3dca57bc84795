"""FrozenBatchNorm2d / get_norm — surface of detectron2/layers/batch_norm.py:14-153.

In the MI355X path a FrozenBatchNorm2d that follows a Conv2d is never run as its own pass: the
Conv2d wrapper reads ``scale_bias()`` and hands it to the convolution's epilogue.  ``forward`` is
kept for stand-alone use and computes the same affine map with plain tensor ops.
"""
import torch
from torch import nn


class FrozenBatchNorm2d(nn.Module):
    """BatchNorm2d with fixed statistics and affine terms: y = x * scale + bias,
    scale = weight * rsqrt(running_var + eps), bias = bias - running_mean * scale."""

    _version = 3

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features = num_features
        self.eps = eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)

    def scale_bias(self):
        """(scale, bias) of the affine map; cached until any of the four buffers is written
        (tensor version counters), so a frozen layer costs no launches per step."""
        key = (self.weight._version, self.bias._version, self.running_mean._version, self.running_var._version,
               self.weight.data_ptr(), self.weight.device)
        cache = self.__dict__.get("_sb_cache")
        if cache is None or cache[0] != key:
            with torch.no_grad():
                scale = self.weight * (self.running_var + self.eps).rsqrt()
                cache = (key, scale, self.bias - self.running_mean * scale)
            self.__dict__["_sb_cache"] = cache
        return cache[1], cache[2]

    def forward(self, x):
        scale, bias = self.scale_bias()
        return x * scale.reshape(1, -1, 1, 1).to(x.dtype) + bias.reshape(1, -1, 1, 1).to(x.dtype)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys,
                              unexpected_keys, error_msgs):
        version = local_metadata.get("version", None)
        if version is None or version < 2:
            # very old checkpoints have no running stats (batch_norm.py:77-85)
            if prefix + "running_mean" not in state_dict:
                state_dict[prefix + "running_mean"] = torch.zeros_like(self.running_mean)
            if prefix + "running_var" not in state_dict:
                state_dict[prefix + "running_var"] = torch.ones_like(self.running_var)
        if version is not None and version < 3:
            # versions < 3 stored var without the eps folded in (batch_norm.py:87-93)
            state_dict[prefix + "running_var"] = state_dict[prefix + "running_var"] - self.eps
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys,
                                      unexpected_keys, error_msgs)

    def __repr__(self):
        return "FrozenBatchNorm2d(num_features=%d, eps=%s)" % (self.num_features, self.eps)

    @classmethod
    def convert_frozen_batchnorm(cls, module):
        """Recursively replace BatchNorm2d/SyncBatchNorm by FrozenBatchNorm2d (same statistics)."""
        bn = (nn.modules.batchnorm.BatchNorm2d, nn.modules.batchnorm.SyncBatchNorm)
        res = module
        if isinstance(module, bn):
            res = cls(module.num_features)
            if module.affine:
                res.weight.data = module.weight.data.clone().detach()
                res.bias.data = module.bias.data.clone().detach()
            res.running_mean.data = module.running_mean.data
            res.running_var.data = module.running_var.data
            res.eps = module.eps
        else:
            for name, child in module.named_children():
                new_child = cls.convert_frozen_batchnorm(child)
                if new_child is not child:
                    res.add_module(name, new_child)
        return res


def get_norm(norm, out_channels):
    """norm: "", "FrozenBN", "GN" or a callable(out_channels) -> module (batch_norm.py:128-153).
    BN/SyncBN/nnSyncBN are outside the JTSM path (every BASELINE config uses FrozenBN / GN / none)."""
    if norm is None:
        return None
    if isinstance(norm, str):
        if len(norm) == 0:
            return None
        table = {"FrozenBN": FrozenBatchNorm2d, "GN": lambda c: nn.GroupNorm(32, c)}
        if norm not in table:
            raise KeyError("jtsm_amd supports norm in %s on the JTSM path, got '%s'" % (sorted(table), norm))
        norm = table[norm]
    return norm(out_channels)
