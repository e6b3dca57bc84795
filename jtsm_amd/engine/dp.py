"""Data-parallel gradient exchange of the training step (the reference wraps its model in torch DDP,
detectron2/engine/defaults.py:288-291,331-336; step: projects/WSL/tools/train_net.py:95-119).

One process per GPU.  Each rank's losses are means over its own images and the ranks' gradients are AVERAGED
(mean of means, exactly what DDP gives the reference).  This module owns the exchange instead of delegating it:

  * at wrap time rank 0's parameters and buffers are broadcast (what DDP's constructor does), so identical weights do
    not rest on every rank having seeded its generator the same way;
  * flat fp32 gradient BUCKETS of at most JTSM_DP_BUCKET_MB (64) MB, one contiguous HBM buffer each.  Every trainable
    parameter's `.grad` is a view into its bucket with the parameter's own strides (channels_last weights included),
    so the fused SGD (solver/build.py) reads the averaged gradients in place and nothing is copied or re-strided after
    the collective.  The first backward runs on a layout guessed from the parameter names (heads, FPN, res5, res4,
    res3); the order in which the gradients ACTUALLY became ready in that backward — rank 0's, broadcast, so every
    rank builds the same layout — then decides the final buckets (a big head bucket no longer waits for the last head
    gradient: fc1's own bucket goes out while the mask heads' backward still runs);
  * the weight-gradient contraction kernels write STRAIGHT into the bucket (layers/conv.py: GRAD_SLOTS), so for the
    convolution / linear weights — 99 % of the bytes — there is no gradient copy at all; the few gradients that
    autograd produces elsewhere (biases, concatenated predictor weights, GroupNorm terms) are copied in by the hook.
    Only the FIRST weight-gradient launch of a parameter in a backward gets the slot: a weight used several times in
    one step (the RPN head runs on five pyramid levels) gets ordinary tensors for its later uses and autograd sums;
  * a post-accumulate hook per parameter counts its bucket down; collectives are issued in BUCKET ORDER on every rank
    (a bucket that completes early waits for its predecessors; the end of the backward issues whatever is left, also
    buckets without any gradient on this rank, zero-filled) — the ranks' collective sequences can never diverge;
  * a bucket's collective runs on a side stream behind an event of the compute stream while the backward continues:
    reduce-scatter (AVG) + all-gather in place — 2 x (N-1)/N x bytes over the xGMI links, each rank reducing 1/N of
    the bucket — or, on backends without reduce-scatter (gloo: the CPU / single-GPU rehearsal), all-reduce;
  * the end of the backward pass (autograd engine callback) makes the compute stream wait for the side stream.

Backend "nccl" is RCCL on ROCm.
"""
import os
import weakref

import torch
import torch.distributed as dist

# the first backward's layout: bucket group i takes the parameters whose name starts with one of these prefixes
# (checked in order; first match wins) — roughly the order the backward reaches them.  Anything unmatched joins the
# first group.  Groups are cut into buckets of at most the cap; the observed order replaces this guess after one step.
BUCKET_PREFIXES = (
    ("roi_heads.", "sem_seg_head.", "proposal_generator."),
    ("backbone.fpn_",),
    ("backbone.bottom_up.res5", "backbone.res5"),
    ("backbone.bottom_up.res4", "backbone.res4"),
    ("backbone.bottom_up.res3", "backbone.res3", "backbone.bottom_up.res2", "backbone.res2", "backbone.bottom_up.stem",
     "backbone.stem"),
)


def bucket_cap_bytes():
    return int(float(os.environ.get("JTSM_DP_BUCKET_MB", "64")) * (1 << 20))


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


def _flat_view(t):
    """The 1-D view over a dense tensor's memory (any permutation of a contiguous layout)."""
    return t.as_strided((t.numel(),), (1,), t.storage_offset())


def broadcast_state(model, src=0, group=None):
    """Rank `src`'s parameters and buffers to every rank (DDP's constructor does the same,
    detectron2/engine/defaults.py:288-291 -> torch DistributedDataParallel._sync_module_states): one flat buffer per
    dtype, one broadcast each."""
    if not _active():
        return 0
    by_dtype = {}
    for t in list(model.parameters()) + list(model.buffers()):
        if t.numel():
            by_dtype.setdefault((t.dtype, t.device), []).append(t.data)
    total = 0
    for (dtype, device), tensors in by_dtype.items():
        dense = [t if _dense(t) else None for t in tensors]
        flat = torch.cat([(_flat_view(t) if d is not None else t.reshape(-1)) for t, d in zip(tensors, dense)])
        dist.broadcast(flat, src=src, group=group)
        off = 0
        for t, d in zip(tensors, dense):
            piece = flat[off:off + t.numel()]
            if d is not None:
                _flat_view(t).copy_(piece)
            else:
                t.copy_(piece.view(t.shape))
            off += t.numel()
        total += flat.numel() * flat.element_size()
    return total


class _Bucket(object):
    __slots__ = ("flat", "params", "pending", "event", "numel")


class GradSlot(object):
    """What layers/conv.py finds for a weight: the bucket view, and whether a launch of THIS backward already wrote it."""
    __slots__ = ("view", "param", "owner", "written")

    def __init__(self, view, param, owner):
        self.view, self.param, self.owner, self.written = view, weakref.ref(param), weakref.ref(owner), False


class GradientExchange(object):
    """Bucketed, overlapped gradient averaging for `model` (see the module docstring).  Works for world size 1 too
    (no collective; gradients still land in the flat buckets), which is how the single-process tests exercise the
    slot / hook machinery."""

    def __init__(self, model, device=None, collective=None, group=None, force_collectives=False, cap_bytes=None,
                 rebucket=True):
        """force_collectives: issue the collectives even in a one-rank group (they are then copies) — how the GPU
        suite drives the RCCL calls and the side-stream ordering on a single device.  rebucket: after the first
        backward lay the buckets out in the observed order of gradient completion."""
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
        self.cap = bucket_cap_bytes() if cap_bytes is None else int(cap_bytes)
        from ..layers import conv
        for n, p in named:
            if not _dense(p):
                raise RuntimeError("GradientExchange: parameter %s is not dense in memory" % n)
            e = conv.GRAD_SLOTS.get((p.data_ptr(), p.numel()))
            if e is not None and e.param() is p and e.owner() is not None:
                raise RuntimeError("GradientExchange: parameter %s already belongs to a live exchange (detach() it "
                                   "first)" % n)
        groups = [[] for _ in BUCKET_PREFIXES]
        for n, p in named:
            idx = next((i for i, pre in enumerate(BUCKET_PREFIXES) if n.startswith(pre)), 0)
            groups[idx].append(p)
        # within a group: roughly the order the backward finishes them
        order = [p for members in groups for p in members[::-1]]
        self._names = {p: n for n, p in named}
        self._index = {p: i for i, (_, p) in enumerate(named)}
        self._by_index = [p for _, p in named]
        self.buckets, self._slot, self._handles = [], {}, []
        self._layout(order, boundaries=[len(g) for g in groups if g])
        ref = weakref.ref(self)   # the hooks must not keep the exchange (and through it the buckets) alive

        def hook(p):
            ex = ref()
            if ex is not None:
                ex._hook(p)
        for p in self._by_index:
            self._handles.append(p.register_post_accumulate_grad_hook(hook))
            # (layers/conv.py: these hooks do not tie a gradient to the compute stream — _launch waits for every stream
            # that produces gradients — so the weight-gradient and semantic-head side streams stay on under the exchange)
            p._jtsm_exchange_waits_for_producers = True
        self.main_stream = None            # the stream the forward ran on (set by the forward pre-hook below)

        def note_stream(module, args):
            ex = ref()
            if ex is not None and ex.cuda:
                ex.main_stream = torch.cuda.current_stream(ex.device)
        self._handles.append(model.register_forward_pre_hook(note_stream))
        self.comm_stream = torch.cuda.Stream(device=self.device) if (self.cuda and (self.world > 1 or self.force)) \
            else None
        self._in_backward = False
        self._next = 0                     # the next bucket (index) whose collective may be issued
        self._seen = []                    # parameters in the order their gradients became ready (first backward)
        self._counted = set()              # parameters whose gradient this backward has counted already
        self._rebucket = bool(rebucket) and os.environ.get("JTSM_DP_REBUCKET", "1") != "0"
        self.rebucketed = False
        self.issue_log = None              # tests: set to [] to record the bucket index of every collective issued

    # ---- bucket layout ------------------------------------------------------------------------------------------
    def _layout(self, order, boundaries=None):
        """Cut `order` into buckets of at most the cap (a parameter above the cap gets a bucket of its own; a
        boundary of the name-based guess also closes a bucket) and point every parameter's slot into them.  Existing
        gradients move along."""
        from ..layers import conv
        cuts, run = set(), 0
        for b in boundaries or []:
            run += b
            cuts.add(run)
        chunks, cur, cur_bytes = [], [], 0
        for i, p in enumerate(order):
            nbytes = 4 * ((p.numel() + 63) // 64 * 64)
            if cur and (cur_bytes + nbytes > self.cap or i in cuts):
                chunks.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            chunks.append(cur)
        old = self._slot
        self.buckets, self._slot = [], {}
        for members in chunks:
            b = _Bucket()
            # pad every slot to 64 floats (256 B) and the bucket to a multiple of 64 * world so that reduce-scatter
            # shards are equal and 256-byte aligned
            offs, total = [], 0
            for p in members:
                offs.append(total)
                total += (p.numel() + 63) // 64 * 64
            total = (total + 64 * self.world - 1) // (64 * self.world) * (64 * self.world)
            b.flat = torch.zeros(total, dtype=torch.float32, device=self.device)
            b.numel = total
            b.params = list(members)
            b.pending = len(members)
            b.event = torch.cuda.Event() if self.cuda else None
            for p, off in zip(members, offs):
                view = b.flat[off:off + p.numel()].as_strided(p.shape, p.stride())
                if p in old:
                    prev = old[p][1]
                    if p.grad is not None and p.grad.data_ptr() == prev.data_ptr():
                        view.copy_(prev)
                        p.grad = view
                self._slot[p] = (b, view)
                conv.GRAD_SLOTS[(p.data_ptr(), p.numel())] = GradSlot(view, p, self)
            self.buckets.append(b)
        self.bytes = 4 * sum(b.numel for b in self.buckets)

    def _adopt_observed_order(self):
        """After the first backward: rank 0's completion order becomes everyone's bucket order."""
        seen, have = [], set()
        for p in self._seen:
            if p not in have:
                have.add(p)
                seen.append(p)
        for b in self.buckets:            # parameters without a gradient in that step keep their old relative place
            for p in b.params:
                if p not in have:
                    have.add(p)
                    seen.append(p)
        idx = torch.tensor([self._index[p] for p in seen], dtype=torch.int64,
                           device=self.device if self.cuda and dist.is_initialized() and
                           dist.get_backend(self.group) == "nccl" else "cpu")
        if self.world > 1:
            dist.broadcast(idx, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                           group=self.group)
        order = [self._by_index[i] for i in idx.tolist()]
        assert sorted(self._index[p] for p in order) == list(range(len(self._by_index)))
        self._layout(order)
        self.rebucketed = True

    # ---- per-parameter hook: runs right after autograd has stored p.grad
    def _hook(self, p):
        from ..layers import conv
        # autograd runs the hook also when a node handed it no gradient — a weight whose gradient is queued for the
        # grouped launch (layers/conv.py) comes again from _deliver_grad.  `.grad is None` alone does not tell: with
        # zero_grad(set_to_none=False), or in a second backward before the optimizer step, `.grad` still holds the
        # previous gradient
        if p.grad is None or conv.deferred_pending(p):
            return
        if p in self._counted:       # once per backward: a second call must not count the bucket down again
            return
        self._counted.add(p)
        b, view = self._slot[p]
        g = p.grad
        if g.data_ptr() != view.data_ptr():
            view.copy_(g)            # produced outside the slot-aware kernels: one small copy
            p.grad = view
        if not self._in_backward:
            self._in_backward = True
            torch.autograd.Variable._execution_engine.queue_callback(self._finish)
        if not self.rebucketed:
            self._seen.append(p)
        b.pending -= 1
        while self._next < len(self.buckets) and self.buckets[self._next].pending == 0:
            self._launch(self._next)
            self._next += 1

    def _launch(self, i):
        b = self.buckets[i]
        if self.issue_log is not None:
            self.issue_log.append(i)
        if self.world == 1 and not self.force:
            return
        if self.comm_stream is None:     # CPU tensors (gloo)
            dist.all_reduce(b.flat, group=self.group)
            b.flat.div_(self.world)
            return
        # The bucket's gradients are complete in STREAM order on the streams that produced them: the current stream (the
        # hook runs under the stream of the node that delivered the last gradient — the compute stream, or a side stream
        # of layers/conv.py / meta_arch/mcnn.py), the compute stream, and every other registered producer stream.  The
        # collective waits for all of them.
        from ..layers import conv
        current = torch.cuda.current_stream(self.device)
        b.event.record(current)
        others = [s for s in [self.main_stream] + list(conv.PRODUCER_STREAMS)
                  if s is not None and s.device == self.device and s != current]
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(b.event)
            for s in others:
                self.comm_stream.wait_stream(s)
            if self.collective == "rs_ag":
                # the in-place forms RCCL documents: recv == send + rank * count (reduce-scatter), send == recv + rank *
                # count (all-gather)
                shard = b.numel // self.world
                r = dist.get_rank(self.group)
                mine = b.flat[r * shard:(r + 1) * shard]
                dist.reduce_scatter_tensor(mine, b.flat, op=dist.ReduceOp.AVG, group=self.group)
                dist.all_gather_into_tensor(b.flat, mine, group=self.group)
            else:
                dist.all_reduce(b.flat, group=self.group)
                b.flat.div_(self.world)

    def _finish(self):
        """End of the backward pass: every bucket not yet issued goes out, IN INDEX ORDER — a bucket some of whose
        parameters got no gradient this step, or none at all, is completed with zeros (another rank may hold real
        gradients for it, and all ranks must issue the same collectives) — then the compute stream waits for the side
        stream."""
        from ..layers import conv
        conv.flush_deferred_weight_gradients()   # queued weight gradients reach their slots (and hooks) before the close
        while self._next < len(self.buckets):
            b = self.buckets[self._next]
            if b.pending:
                for p in b.params:   # a parameter that received no gradient contributes zeros
                    if p.grad is None:
                        self._slot[p][1].zero_()
                        p.grad = self._slot[p][1]
            self._launch(self._next)
            self._next += 1
        for b in self.buckets:
            b.pending = len(b.params)
        self._next = 0
        self._counted = set()
        if self.comm_stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
        for p in self._slot:
            e = conv.GRAD_SLOTS.get((p.data_ptr(), p.numel()))
            if e is not None:
                e.written = False
        self._in_backward = False
        if self._rebucket and not self.rebucketed:
            self._adopt_observed_order()
        self._seen = []

    def detach(self):
        """Forget the kernel-side slots and the hooks (tests that build several exchanges in one process)."""
        from ..layers import conv
        for p in self._slot:
            e = conv.GRAD_SLOTS.get((p.data_ptr(), p.numel()))
            if e is not None and e.owner() is self:
                conv.GRAD_SLOTS.pop((p.data_ptr(), p.numel()), None)
        for h in self._handles:
            h.remove()
        self._handles = []
        for p in self._by_index:
            if hasattr(p, "_jtsm_exchange_waits_for_producers"):
                del p._jtsm_exchange_waits_for_producers


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


def wrap_data_parallel(model, device=None, collective=None, broadcast=True):
    """The model behind this repo's own gradient exchange (identity for a single process).  As DDP does in the
    reference (detectron2/engine/defaults.py:288-291), rank 0's parameters and buffers replace every other rank's."""
    if not _active():
        return model
    if broadcast:
        broadcast_state(model)
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
