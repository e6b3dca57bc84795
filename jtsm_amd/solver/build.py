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

    # ---- the heads' share of the step, early (opt-in: attach_early_heads) ---------------------------------------------
    def attach_early_heads(self, model, prefixes=("roi_heads.", "sem_seg_head.")):
        """Update the parameters under `prefixes` as soon as their gradients are complete — when the backward pass reaches
        the FPN (layers/conv.py: HEADS_DONE_HOOKS) — on the weight-gradient side stream, beside the FPN's and the backbone's
        backward, and refresh their operand planes there too: for the R50-FPN composite that is 70 % of the step's 0.25 ms
        update and of the next step's 0.26 ms plane refresh, both bound by HBM bytes, moved from the serial end of the step
        to beside latency-bound launches.  `step()` then updates the rest.  Same arithmetic, same bits.  ONLY for loops that
        call step() after every backward (no gradient accumulation over several backward passes), and not under
        post-accumulate hooks (the gradient exchange averages first): both are checked at run time where they can be."""
        import os
        import weakref
        from ..layers import conv
        if os.environ.get("JTSM_EARLY_STEP", "1") == "0":
            return self
        self._early_ids = {id(p) for n, p in model.named_parameters() if p.requires_grad and n.startswith(tuple(prefixes))}
        self._stepped = set()
        ref = weakref.ref(self)

        def hook():
            opt = ref()
            if opt is not None:
                opt._early_step()
        self._early_hook = hook
        conv.HEADS_DONE_HOOKS.append(hook)
        return self

    def detach_early_heads(self):
        from ..layers import conv
        hook = self.__dict__.pop("_early_hook", None)
        if hook in conv.HEADS_DONE_HOOKS:
            conv.HEADS_DONE_HOOKS.remove(hook)
        self._early_ids = set()

    @torch.no_grad()
    def _early_step(self):
        from ..layers import conv
        ids = getattr(self, "_early_ids", None)
        if not ids or not conv.WGRAD_STREAM or self._stepped:
            return
        todo = []
        for group in self.param_groups:
            for p in group["params"]:
                if id(p) in ids and p.grad is not None:
                    if not p.is_cuda or getattr(p, "_post_accumulate_grad_hooks", None):
                        return                      # (under the gradient exchange the step waits for the averaged gradients)
                    todo.append(p)
        # the heads' queued weight gradients (the mask towers' group) are delivered before anything is updated
        conv.flush_deferred_weight_gradients()
        todo = [p for g in self.param_groups for p in g["params"] if id(p) in ids and p.grad is not None]
        if not todo or not conv.queue_side_stream_join():
            return
        device = todo[0].device
        side = conv._wgrad_side_stream(device)
        # every kernel that reads these weights (or their planes) or writes these gradients has been enqueued: the heads'
        # nodes have all run (layers/conv.py: _heads_done) — on the compute stream, the semantic head's stream, this one
        side.wait_stream(torch.cuda.current_stream(device))
        for s in conv.PRODUCER_STREAMS:
            if s is not side and s.device == device:
                side.wait_stream(s)
        with torch.cuda.stream(side):
            for p in todo:
                p.grad.record_stream(side)          # (freed by zero_grad on the host while this stream may still read it)
            self._apply(only=ids, staging="_staging_early")
            conv.refresh_weight_planes()            # the updated weights' operand planes, beside the backward as well
        self._stepped = {id(p) for p in todo}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._apply(skip=getattr(self, "_stepped", None) or None)
        if getattr(self, "_stepped", None):
            self._stepped = set()
        return loss

    def _apply(self, only=None, skip=None, staging="_staging"):
        rows, blocks, device, keep, touched = [], 0, None, [], []
        for group in self.param_groups:
            lr, wd, mu = _f32_bits(group["lr"]), _f32_bits(group["weight_decay"]), _f32_bits(group["momentum"])
            for p in group["params"]:
                g = p.grad
                if g is None or (only is not None and id(p) not in only) or (skip is not None and id(p) in skip):
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
            st = self.__dict__.setdefault(staging, {"host": [None, None], "dev": None, "turn": 0,
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
