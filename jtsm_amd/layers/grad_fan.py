"""Gradient fan-in without autograd's additions.

A tensor with several consumers (an FPN level read by the box pooler, the mask pooler and the semantic head; a
res-stage output read by the next stage and by the FPN lateral) gets its gradient as `g1 + g2 + g3`: every consumer
writes a full map and autograd adds them pairwise — two passes over the level's largest maps per extra consumer
(0.34 ms per step).  Here the consumers are this package's own autograd nodes, whose backward kernels can ADD INTO an
existing map for free (a convolution's data-gradient epilogue has a residual operand; the pooling gathers own every
cell they write).  So:

    a, b, c = fan_out(x, 3)          # three views of x, one per consumer, sharing a FanRecord

Each consumer that understands the protocol *claims* its view in forward (`claim(view)`), and in backward either
finds the record empty — it computes its gradient the ordinary way and leaves it IN THE RECORD (`offer`) — or finds a
map there and adds its own term into it in place; either way it returns None to autograd.  `_Fan.backward`, which
runs after every consumer, returns the record's map plus whatever arrived through autograd itself: the gradient of
a consumer that does not take part (an ordinary tensor op on a view, a second reader of a claimed view) lands in that
view's slot and is added the ordinary way.  Correctness therefore does not depend on who takes part: the shared map
never passes through autograd's own accumulation, so nothing can copy it half-way and miss a later addition.

The sum's order is the backward's node order — fixed for a fixed graph, so results stay reproducible run to run —
rather than autograd's (g1 + g2) + g3.
"""
import os

import torch
from torch.autograd import Function

ENABLED = os.environ.get("JTSM_GRAD_FAN", "1") != "0"     # (A/B switch)
_ATTR, _CLAIMED = "_jtsm_fan", "_jtsm_fan_claimed"
STATS = {"nodes": 0, "slots_filled": 0}      # (tests: a fan whose consumers all took part leaves every slot empty)


class FanRecord:
    __slots__ = ("buffer",)

    def __init__(self):
        self.buffer = None      # the gradient map the consumers of this tensor accumulate into (this backward pass)


class _Fan(Function):
    @staticmethod
    def forward(ctx, x, n, rec):
        ctx.rec = rec
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        total, ctx.rec.buffer = ctx.rec.buffer, None   # (a later backward through the same graph starts clean)
        STATS["nodes"] += 1
        for g in grads:
            if g is not None:
                STATS["slots_filled"] += 1
                total = g if total is None else total + g
        return total, None, None


def fan_out(x, n):
    """n views of x for n consumers (see the module docstring).  Without gradients, or switched off: x itself."""
    if not (ENABLED and n > 1 and torch.is_grad_enabled() and x.requires_grad):
        return (x,) * n
    rec = FanRecord()
    outs = _Fan.apply(x, n, rec)
    for o in outs:
        setattr(o, _ATTR, rec)
    return outs


def claim(x):
    """Called by a participating node's forward on its INPUT tensor object: the record to use in backward, or None
    (not a fan view, or a view somebody claimed already)."""
    rec = getattr(x, _ATTR, None)
    if rec is None or getattr(x, _CLAIMED, False):
        return None
    setattr(x, _CLAIMED, True)
    return rec


def target(rec, shape, device):
    """In backward: the map to add into, or None when this consumer is the first (or does not take part)."""
    if rec is None:
        return None
    b = rec.buffer
    if (b is not None and tuple(b.shape) == tuple(shape) and b.dtype == torch.float32 and b.device == device and
            b.is_contiguous(memory_format=torch.channels_last)):
        return b
    return None


def offer(rec, grad):
    """In backward, by the first consumer: leave the freshly written gradient map in the record for the others to add
    into.  True: taken — return None to autograd for this input; False: hand `grad` to autograd as usual."""
    if rec is not None and rec.buffer is None and grad is not None and grad.dtype == torch.float32 and \
            grad.dim() == 4 and grad.is_contiguous(memory_format=torch.channels_last):
        rec.buffer = grad
        return True
    return False
