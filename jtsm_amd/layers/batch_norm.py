"""FrozenBatchNorm2d / get_norm — the API of detectron2/layers/batch_norm.py:14-153 (class name, buffer names and
state-dict versioning are fixed by the checkpoints; everything else is this repo's).

On the MI355X path a frozen norm behind a convolution never runs as its own pass: `Conv2d` asks for `scale_bias()`
— the affine map y = x * scale + bias with scale = weight / sqrt(running_var + eps), bias = bias - running_mean * scale
— and the contraction's epilogue applies it.  The pair is memoised on the buffers' version counters, so a frozen layer
costs no launch per step.  `forward` exists for stand-alone use and applies the same map with tensor ops.
"""
import torch
from torch import nn

_STATS = ("weight", "bias", "running_mean", "running_var")


class FrozenBatchNorm2d(nn.Module):
    _version = 3          # state-dict layout version, as the reference writes it into checkpoints' metadata

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps
        start = {"weight": 1.0, "bias": 0.0, "running_mean": 0.0, "running_var": 1.0 - eps}
        for name in _STATS:
            self.register_buffer(name, torch.full((num_features,), start[name]))
        self._memo = None

    # ---- the affine map -------------------------------------------------------------------------------------------
    def _stamp(self):
        return tuple(getattr(self, n)._version for n in _STATS) + (self.weight.data_ptr(), str(self.weight.device))

    def scale_bias(self):
        """(scale, bias), each (C,), recomputed only after one of the four buffers has been written or moved."""
        stamp = self._stamp()
        if self._memo is None or self._memo[0] != stamp:
            with torch.no_grad():
                scale = self.weight * torch.rsqrt(self.running_var + self.eps)
                self._memo = (stamp, scale, self.bias - self.running_mean * scale)
        return self._memo[1], self._memo[2]

    def forward(self, x):
        scale, bias = (t.to(x.dtype).view(1, -1, 1, 1) for t in self.scale_bias())
        return x * scale + bias

    def extra_repr(self):
        return "num_features=%d, eps=%s" % (self.num_features, self.eps)

    # ---- checkpoints of older layouts (batch_norm.py:77-93) ---------------------------------------------------------
    def _load_from_state_dict(self, state_dict, prefix, local_metadata, *rest):
        stored = local_metadata.get("version")
        if stored is None or stored < 2:        # no running statistics in the file: identity statistics
            state_dict.setdefault(prefix + "running_mean", torch.zeros_like(self.running_mean))
            state_dict.setdefault(prefix + "running_var", torch.ones_like(self.running_var))
        if stored is not None and stored < 3:   # the variance was stored WITH eps
            key = prefix + "running_var"
            state_dict[key] = state_dict[key] - self.eps
        super()._load_from_state_dict(state_dict, prefix, local_metadata, *rest)

    # ---- freezing a trained BatchNorm ---------------------------------------------------------------------------------
    @classmethod
    def from_batchnorm(cls, bn):
        frozen = cls(bn.num_features, bn.eps)
        with torch.no_grad():
            if bn.affine:
                frozen.weight.copy_(bn.weight)
                frozen.bias.copy_(bn.bias)
            frozen.running_mean.copy_(bn.running_mean)
            frozen.running_var.copy_(bn.running_var)
        return frozen.to(bn.running_mean.device)

    @classmethod
    def convert_frozen_batchnorm(cls, module):
        """`module` with every BatchNorm2d / SyncBatchNorm in it (or `module` itself) replaced by its frozen form."""
        live = (nn.BatchNorm2d, nn.SyncBatchNorm)
        if isinstance(module, live):
            return cls.from_batchnorm(module)
        for parent in list(module.modules()):
            for name, child in list(parent.named_children()):
                if isinstance(child, live):
                    setattr(parent, name, cls.from_batchnorm(child))
        return module


_NORMS = {"FrozenBN": FrozenBatchNorm2d, "GN": lambda channels: nn.GroupNorm(32, channels)}


def get_norm(norm, out_channels):
    """None / "" -> no norm; "FrozenBN" | "GN" (what the BASELINE configs use) or any callable(channels) -> module.
    The trainable batch norms of batch_norm.py:128-153 (BN, SyncBN, nnSyncBN) are outside the JTSM path."""
    if not norm:
        return None
    if callable(norm):
        return norm(out_channels)
    if norm not in _NORMS:
        raise KeyError("jtsm_amd supports norm in %s on the JTSM path, got '%s'" % (sorted(_NORMS), norm))
    return _NORMS[norm](out_channels)
