"""Optimizer of the hot path: SGD with momentum exactly as detectron2/solver/build.py:110-195 builds it for the
JTSM configs (torch.optim.SGD, per-parameter lr / weight decay: BIAS_LR_FACTOR, WEIGHT_DECAY_BIAS,
WEIGHT_DECAY_NORM), applied to every parameter in ONE launch of libjtsm_hip.so (csrc/elementwise.hip:
sgd_multi_kernel) instead of torch's three multi-tensor passes."""
import ctypes as C
import struct

import torch

from .. import _lib as L


def _f32_bits(x):
    return struct.unpack("<I", struct.pack("<f", float(x)))[0]


class SGD(torch.optim.Optimizer):
    """torch.optim.SGD(params, lr, momentum, weight_decay) semantics (dampening 0, nesterov False); state_dict
    compatible (`momentum_buffer` per parameter)."""

    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0):
        if momentum <= 0.0:
            raise ValueError("jtsm_amd SGD is the momentum form (the reference trains with momentum 0.9)")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        rows, blocks, device, keep, touched = [], 0, None, [], []
        for group in self.param_groups:
            lr, wd, mu = _f32_bits(group["lr"]), _f32_bits(group["weight_decay"]), _f32_bits(group["momentum"])
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32:
                    raise RuntimeError("jtsm_amd SGD updates float32 parameters on the HIP device only")
                st = self.state[p]
                if "momentum_buffer" not in st:
                    # a parameter's first gradient may come at any step (a layer unfrozen later, a head whose loss was
                    # skipped): start from a zero buffer — mu * 0 + d is exactly torch.optim.SGD's first step (buf = d)
                    st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                buf = st["momentum_buffer"]
                if g.stride() != p.stride() or buf.stride() != p.stride():
                    # the kernel walks p, grad and the momentum buffer in the same element order
                    g = torch.empty_like(p, memory_format=torch.preserve_format).copy_(g)
                    keep.append(g)
                n = p.numel()
                rows.append([p.data_ptr(), g.data_ptr(), buf.data_ptr(), n, blocks, lr, wd, mu])
                touched.append(p)
                touched.append(buf)
                blocks += (n + 1023) // 1024
                device = p.device
        if rows:
            # table upload without stalling the host: two alternating pinned staging buffers and an asynchronous copy on
            # the launch stream; a buffer is rewritten only after the copy that last read it has finished (event)
            nwords = len(rows) * 8
            st = self.__dict__.setdefault("_staging", {"host": [None, None], "dev": None, "turn": 0,
                                                       "copied": [None, None]})
            if st["dev"] is None or st["dev"].numel() < nwords or st["dev"].device != device:
                st["host"] = [torch.empty(nwords, dtype=torch.int64).pin_memory() for _ in range(2)]
                st["dev"] = torch.empty(nwords, dtype=torch.int64, device=device)
                st["copied"] = [None, None]
            st["turn"] ^= 1
            host = st["host"][st["turn"]]
            if st["copied"][st["turn"]] is not None:
                st["copied"][st["turn"]].synchronize()
            host[:nwords].copy_(torch.tensor(rows, dtype=torch.int64).view(-1))
            table = st["dev"][:nwords]
            table.copy_(host[:nwords], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            st["copied"][st["turn"]] = ev
            L.note_bytes(20.0 * sum(r[3] for r in rows))   # param, grad, momentum read; param, momentum written
            L.check(L.lib().jtsm_sgd_momentum_multi_f32(L.ptr(table), len(rows), C.c_long(blocks), 0, L.stream()),
                    "sgd_momentum_multi")
            torch.autograd.graph.increment_version(touched)   # updated behind autograd's back: say so
        return loss


def build_optimizer(cfg, model):
    """detectron2/solver/build.py:110-195 for SOLVER.OPTIMIZER-less configs: SGD, bias lr x BIAS_LR_FACTOR,
    WEIGHT_DECAY_BIAS for biases, WEIGHT_DECAY_NORM for normalisation layers, WEIGHT_DECAY otherwise."""
    norm_types = (torch.nn.BatchNorm2d, torch.nn.GroupNorm, torch.nn.LayerNorm, torch.nn.SyncBatchNorm)
    groups, seen = [], set()
    for module in model.modules():
        for name, p in module.named_parameters(recurse=False):
            if not p.requires_grad or p in seen:
                continue
            seen.add(p)
            lr, wd = cfg.SOLVER.BASE_LR, cfg.SOLVER.WEIGHT_DECAY
            if isinstance(module, norm_types):
                wd = cfg.SOLVER.WEIGHT_DECAY_NORM
            elif name == "bias":
                lr = cfg.SOLVER.BASE_LR * cfg.SOLVER.BIAS_LR_FACTOR
                wd = cfg.SOLVER.WEIGHT_DECAY_BIAS
            groups.append({"params": [p], "lr": lr, "weight_decay": wd})
    return SGD(groups, cfg.SOLVER.BASE_LR, momentum=cfg.SOLVER.MOMENTUM)
