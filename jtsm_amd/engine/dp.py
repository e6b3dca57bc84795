"""Data-parallel gradient exchange of the training step (the reference wraps its model in torch DDP,
detectron2/engine/defaults.py:288-291; step: projects/WSL/tools/train_net.py:95-119).

One process per GPU.  Each rank's losses are means over its own images and the ranks' gradients are AVERAGED
(mean of means, exactly what DDP gives the reference).  This module owns the exchange instead of delegating it:

  * flat fp32 gradient BUCKETS laid out in the order the backward produces them — heads, FPN, res5, res4, res3 — one
    contiguous HBM buffer per bucket.  Every trainable parameter's `.grad` is a view into its bucket with the
    parameter's own strides (channels_last weights included), so the fused SGD (solver/build.py) reads the averaged
    gradients in place and nothing is copied or re-strided after the collective;
  * the weight-gradient contraction kernels write STRAIGHT into the bucket (layers/conv.py: GRAD_SLOTS), so for the
    convolution / linear weights — 99 % of the bytes — there is no gradient copy at all; the few gradients that
    autograd produces elsewhere (biases, concatenated predictor weights, GroupNorm terms) are copied in by the hook;
  * a post-accumulate hook per parameter counts its bucket down; when a bucket is complete the compute stream
    records an event and the bucket's collective starts on a side stream while the backward continues:
    reduce-scatter (AVG) + all-gather in place — 2 x (N-1)/N x bytes over the xGMI links, each rank reducing 1/N of
    the bucket — or, on backends without reduce-scatter (gloo: the CPU / single-GPU rehearsal), all-reduce;
  * the end of the backward pass (autograd engine callback) makes the compute stream wait for the side stream.

Backend "nccl" is RCCL on ROCm.
"""
import os

import torch
import torch.distributed as dist

# bucket i takes the parameters whose name starts with one of these prefixes (checked in order; first match wins):
# the order the backward reaches them.  Anything unmatched joins the first bucket.
BUCKET_PREFIXES = (
    ("roi_heads.", "sem_seg_head.", "proposal_generator."),
    ("backbone.fpn_",),
    ("backbone.bottom_up.res5", "backbone.res5"),
    ("backbone.bottom_up.res4", "backbone.res4"),
    ("backbone.bottom_up.res3", "backbone.res3", "backbone.bottom_up.res2", "backbone.res2", "backbone.bottom_up.stem",
     "backbone.stem"),
)


def env_ranks():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_distributed(backend, device=None):
    rank, world, _ = env_ranks()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def _active():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class _Bucket(object):
    __slots__ = ("flat", "params", "pending", "event", "numel")


class GradientExchange(object):
    """Bucketed, overlapped gradient averaging for `model` (see the module docstring).  Works for world size 1 too
    (no collective; gradients still land in the flat buckets), which is how the single-process tests exercise the
    slot / hook machinery."""

    def __init__(self, model, device=None, collective=None, group=None, force_collectives=False):
        """force_collectives: issue the collectives even in a one-rank group (they are then copies) — how the GPU
        suite drives the RCCL calls and the side-stream ordering on a single device."""
        self.model, self.group = model, group
        self.force = bool(force_collectives) and dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if (_active() or self.force) else 1
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        if not named:
            raise ValueError("GradientExchange: the model has no trainable parameter")
        self.device = device if device is not None else named[0][1].device
        self.cuda = self.device.type == "cuda"
        backend = dist.get_backend(group) if (self.world > 1 or self.force) else None
        if collective is None:
            collective = os.environ.get("JTSM_DP_COLLECTIVE", "rs_ag" if backend == "nccl" else "allreduce")
        if collective not in ("rs_ag", "allreduce"):
            raise ValueError("collective must be 'rs_ag' or 'allreduce'")
        self.collective = collective
        groups = [[] for _ in BUCKET_PREFIXES]
        for n, p in named:
            if not _dense(p):
                raise RuntimeError("GradientExchange: parameter %s is not dense in memory" % n)
            idx = next((i for i, pre in enumerate(BUCKET_PREFIXES) if n.startswith(pre)), 0)
            groups[idx].append((n, p))
        self.buckets, self._slot = [], {}
        from ..layers import conv
        for members in groups:
            if not members:
                continue
            members = members[::-1]   # within a bucket: roughly the order the backward finishes them
            b = _Bucket()
            # pad every slot to 64 floats (256 B) and the bucket to a multiple of 64 * world so that reduce-scatter
            # shards are equal and 256-byte aligned
            offs, total = [], 0
            for _, p in members:
                offs.append(total)
                total += (p.numel() + 63) // 64 * 64
            total = (total + 64 * self.world - 1) // (64 * self.world) * (64 * self.world)
            b.flat = torch.zeros(total, dtype=torch.float32, device=self.device)
            b.numel = total
            b.params = [p for _, p in members]
            b.pending = len(members)
            b.event = torch.cuda.Event() if self.cuda else None
            for (n, p), off in zip(members, offs):
                view = b.flat[off:off + p.numel()].as_strided(p.shape, p.stride())
                self._slot[p] = (b, view)
                conv.GRAD_SLOTS[(p.data_ptr(), p.numel())] = view
                p.register_post_accumulate_grad_hook(self._hook)
            self.buckets.append(b)
        self.comm_stream = torch.cuda.Stream(device=self.device) if (self.cuda and (self.world > 1 or self.force)) \
            else None
        self._in_backward = False
        self.bytes = 4 * sum(b.numel for b in self.buckets)

    # ---- per-parameter hook: runs right after autograd has stored p.grad
    def _hook(self, p):
        b, view = self._slot[p]
        g = p.grad
        if g.data_ptr() != view.data_ptr():
            view.copy_(g)            # produced outside the slot-aware kernels: one small copy
            p.grad = view
        if not self._in_backward:
            self._in_backward = True
            torch.autograd.Variable._execution_engine.queue_callback(self._finish)
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def _launch(self, b):
        if self.world == 1 and not self.force:
            return
        if self.comm_stream is None:     # CPU tensors (gloo)
            dist.all_reduce(b.flat, group=self.group)
            b.flat.div_(self.world)
            return
        b.event.record()                 # on the compute stream: the bucket's gradients are complete here
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(b.event)
            if self.collective == "rs_ag":
                shard = b.numel // self.world
                mine = b.flat[dist.get_rank(self.group) * shard:(dist.get_rank(self.group) + 1) * shard]
                dist.reduce_scatter_tensor(mine, b.flat, op=dist.ReduceOp.AVG, group=self.group)
                dist.all_gather_into_tensor(b.flat, mine, group=self.group)
            else:
                dist.all_reduce(b.flat, group=self.group)
                b.flat.div_(self.world)

    def _finish(self):
        """End of the backward pass: late buckets (a parameter without gradient this step) still go out, then the
        compute stream waits for every collective."""
        for b in self.buckets:
            if 0 < b.pending < len(b.params):
                for p in b.params:   # a parameter that received no gradient contributes zeros
                    if p.grad is None or p.grad.data_ptr() != self._slot[p][1].data_ptr():
                        if p.grad is None:
                            self._slot[p][1].zero_()
                            p.grad = self._slot[p][1]
                self._launch(b)
            b.pending = len(b.params)
        if self.comm_stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
        self._in_backward = False

    def detach(self):
        """Forget the kernel-side slots (tests that build several models in one process)."""
        from ..layers import conv
        for p in self._slot:
            conv.GRAD_SLOTS.pop((p.data_ptr(), p.numel()), None)


def _dense(p):
    """Non-overlapping and dense in memory (any permutation of a contiguous layout)."""
    sizes_strides = sorted(((st, sz) for sz, st in zip(p.shape, p.stride()) if sz > 1))
    expect = 1
    for st, sz in sizes_strides:
        if st != expect:
            return False
        expect *= sz
    return True


class DataParallel(torch.nn.Module):
    """`model` plus its GradientExchange: call it like the model; `backward()` of the losses triggers the exchange."""

    def __init__(self, model, exchange):
        super().__init__()
        self.module = model
        self.exchange = exchange

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


def wrap_data_parallel(model, device=None, collective=None):
    """The model behind this repo's own gradient exchange (identity for a single process)."""
    if not _active():
        return model
    return DataParallel(model, GradientExchange(model, device, collective))


def max_over_ranks(seconds, device):
    if not _active():
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def fence(device=None):
    if _active():
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize()
