"""CPU suite: the N>1 plumbing of bench.py (jtsm_amd/engine/dp.py) with world_size 2 on gloo.
The HIP model itself cannot run without a GPU (tests/test_hip_dp.py runs it, two ranks, on the GPU box), so a small
torch module stands in for it here: what is checked is the data-parallel contract of SURVEY §8e on this repo's own
exchange (flat buckets, per-parameter hooks, end-of-backward callback) — per-rank batches differ (seed 1234 + rank),
every rank ends the step with identical gradients living in the flat buckets, and the applied gradient is the MEAN
over ranks of the per-rank mean losses' gradients."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from jtsm_amd.engine import dp

    r, w = dp.init_distributed("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)                                    # same weights everywhere, like bench.build()
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))
    net = dp.wrap_data_parallel(model)
    g = torch.Generator().manual_seed(1234 + rank)          # per-rank shard of the synthetic data
    x, y = torch.randn(2, 8, generator=g), torch.randn(2, 3, generator=g)
    loss = ((net(x) - y) ** 2).mean()                       # mean over THIS rank's 2 samples
    loss.backward()
    assert isinstance(net, dp.DataParallel) and net.exchange.collective == "allreduce"   # gloo has no reduce-scatter
    for p in model.parameters():     # gradients are views into the flat bucket, with the parameter's strides
        view = net.exchange._slot[p][1]
        assert p.grad.data_ptr() == view.data_ptr() and p.grad.stride() == p.stride()
    # a second step reuses the buckets: hooks, counters and the end-of-backward callback re-arm
    first = torch.cat([p.grad.flatten() for p in model.parameters()]).clone()
    model.zero_grad(set_to_none=True)
    ((net(x) - y) ** 2).mean().backward()
    assert torch.equal(first, torch.cat([p.grad.flatten() for p in model.parameters()]))
    grads = torch.cat([p.grad.flatten() for p in model.parameters()])
    t = dp.max_over_ranks(float(rank + 1), torch.device("cpu"))
    dp.fence()
    out[rank] = (x, y, grads, t)
    torch.distributed.destroy_process_group()


def test_world2_gloo_gradients_are_mean_of_rank_means():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    (x0, y0, g0, t0), (x1, y1, g1, t1) = out[0], out[1]
    assert not torch.equal(x0, x1)                          # different shards
    assert torch.equal(g0, g1)                              # identical averaged gradients on both ranks
    assert t0 == t1 == 2.0                                  # max over ranks
    torch.manual_seed(0)
    ref = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))
    tot = 0.5 * (((ref(x0) - y0) ** 2).mean() + ((ref(x1) - y1) ** 2).mean())
    tot.backward()
    expect = torch.cat([p.grad.flatten() for p in ref.parameters()])
    assert torch.allclose(g0, expect, atol=1e-6)


class _ThreeHeads(torch.nn.Module):
    """A trunk with three heads; `skip` leaves one head out of the loss (its parameters get no gradient)."""

    def __init__(self):
        super().__init__()
        self.trunk = torch.nn.Linear(8, 16)
        self.heads = torch.nn.ModuleList([torch.nn.Linear(16, 3) for _ in range(3)])

    def forward(self, x, skip=None):
        h = torch.relu(self.trunk(x))
        return sum((head(h) ** 2).mean() for i, head in enumerate(self.heads) if i != skip)


def _worker_uneven(rank, world, port, out):
    """Ranks start from DIFFERENT seeds (the wrap broadcasts rank 0's state); on the second step rank 1's loss omits a
    head, on the third no rank touches it: collectives must still be issued in the same order on both ranks, the
    missing gradients count as zeros, and the ranks end identical."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), JTSM_DP_BUCKET_MB="0.0001")     # ~100-byte cap: every tensor its own bucket
    from jtsm_amd.engine import dp

    dp.init_distributed("gloo")
    torch.manual_seed(100 + rank)                            # different weights per rank before the wrap
    model = _ThreeHeads()
    before = torch.cat([p.detach().flatten() for p in model.parameters()]).clone()
    net = dp.wrap_data_parallel(model)
    after = torch.cat([p.detach().flatten() for p in model.parameters()]).clone()
    ex = net.exchange
    assert len(ex.buckets) == 8 and not ex.rebucketed
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.randn(4, 8, generator=g)
    logs, grads = [], []
    for step, skip in enumerate([None, (1 if rank == 1 else None), 1]):
        model.zero_grad(set_to_none=True)
        ex.issue_log = []
        net(x, skip=skip).backward()
        logs.append(list(ex.issue_log))
        grads.append(torch.cat([p.grad.flatten() for p in model.parameters()]).clone())
        assert ex.rebucketed                                 # the observed order replaced the name-based guess
        for p in model.parameters():
            assert p.grad.data_ptr() == ex._slot[p][1].data_ptr()
    order = [ex._names[b.params[0]] for b in ex.buckets]
    out[rank] = (before, after, x, logs, grads, order)
    torch.distributed.destroy_process_group()


def test_world2_broadcast_fixed_issue_order_and_missing_gradients():
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_worker_uneven, args=(world, port, out), nprocs=world, join=True)
    (b0, a0, x0, logs0, g0, order0), (b1, a1, x1, logs1, g1, order1) = out[0], out[1]
    assert not torch.equal(b0, b1)                           # different seeds ...
    assert torch.equal(a0, a1) and torch.equal(a0, b0)       # ... rank 0's state everywhere after the wrap
    assert order0 == order1                                  # same bucket layout (rank 0's observed order)
    # the heads finish before the trunk in a backward: the rebuilt layout starts with head parameters
    assert order0[0].startswith("heads.") and order0[-1].startswith("trunk.")
    for l0, l1 in zip(logs0, logs1):
        assert l0 == l1 == sorted(l0) and len(l0) == 8       # every bucket, in index order, on both ranks, every step
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)
    # step 2: rank 1 had no gradient for head 1 -> the mean is half of rank 0's; step 3: nobody had one -> zeros
    torch.manual_seed(100)
    ref = _ThreeHeads()
    names = [n for n, _ in ref.named_parameters()]
    sizes = [p.numel() for p in ref.parameters()]

    def expect(skips):
        tot = []
        for x, skip in zip((x0, x1), skips):
            ref.zero_grad(set_to_none=True)
            ref(x, skip=skip).backward()
            tot.append(torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).flatten()
                                  for p in ref.parameters()]))
        return 0.5 * (tot[0] + tot[1])

    for got, skips in zip(g0, [(None, None), (None, 1), (1, 1)]):
        assert torch.allclose(got, expect(skips), atol=1e-6)
    off = sum(sizes[:names.index("heads.1.weight")])
    assert g0[2][off:off + sizes[names.index("heads.1.weight")]].abs().max() == 0


class _SlotWgrad(torch.autograd.Function):
    """The kernels' protocol on CPU: the weight gradient of y = x @ w.T is WRITTEN (not accumulated) into the slot
    layers/conv.py hands out, exactly like a fresh weight-gradient launch."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w.t()

    @staticmethod
    def backward(ctx, dy):
        from jtsm_amd.layers import conv
        x, w = ctx.saved_tensors
        dw = dy.t() @ x
        slot = conv.grad_slot(w)
        if slot is not None:
            slot.copy_(dw)
            dw = slot
        return dy @ w, dw


def test_weight_used_twice_in_one_backward_gets_the_sum():
    """a weight applied several times in one step (the RPN head on five pyramid levels) must not
    have its slot overwritten by each use; only the first launch gets the slot, autograd accumulates the rest."""
    from jtsm_amd.engine import dp
    from jtsm_amd.layers import conv

    torch.manual_seed(0)
    model = torch.nn.Linear(4, 3, bias=False)
    ex = dp.GradientExchange(model, torch.device("cpu"))
    try:
        xs = [torch.randn(5, 4) for _ in range(3)]
        for _ in range(2):                                   # the per-backward flags re-arm
            model.zero_grad(set_to_none=True)
            sum(_SlotWgrad.apply(x, model.weight).sum() for x in xs).backward()
            want = sum(torch.ones(5, 3).t() @ x for x in xs)
            assert torch.allclose(model.weight.grad, want, atol=1e-6)
            assert model.weight.grad.data_ptr() == ex._slot[model.weight][1].data_ptr()
        # a second exchange over the same parameters would fight for the slots: refused while the first is alive
        try:
            dp.GradientExchange(model, torch.device("cpu"))
            raise AssertionError("a second live exchange was accepted")
        except RuntimeError as e:
            assert "already belongs" in str(e)
    finally:
        ex.detach()
    assert not any(e.owner() is ex for e in conv.GRAD_SLOTS.values())
    # stale entries (their parameter or exchange is gone) are ignored and dropped
    model2 = torch.nn.Linear(4, 3, bias=False)
    ex2 = dp.GradientExchange(model2, torch.device("cpu"))
    key = (model2.weight.data_ptr(), model2.weight.numel())
    assert key in conv.GRAD_SLOTS
    w_alias = model2.weight.detach()
    del ex2, model2
    import gc
    gc.collect()
    assert conv.grad_slot(w_alias) is None and key not in conv.GRAD_SLOTS


# ---- deferred (grouped) weight gradients under the exchange (both medium items) ---------------------------
class _Planes(object):
    """Stand-in for layers/conv.py's PlaneTensor on the CPU: a 4-d shape and the tensor."""

    def __init__(self, t):
        self.t, self.shape = t, (t.shape[0], t.shape[1], 1, 1)


class _QueuedWgrad(torch.autograd.Function):
    """y = x @ w.T whose weight gradient goes through layers/conv.py's deferral queue exactly as the fused bottleneck
    node's does: the node hands autograd None, the flush delivers the gradient and runs the parameter's hooks."""
    fail = False

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w.flatten(1).t()

    @staticmethod
    def backward(ctx, dy):
        from jtsm_amd.layers import conv
        x, w = ctx.saved_tensors
        dw = conv.planes_backward_weight_deferred(_Planes(dy), _Planes(x), w)
        if _QueuedWgrad.fail:
            raise RuntimeError("backward aborted behind a queued weight gradient")
        return dy @ w.flatten(1), dw


class _TwoLayers(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w1 = torch.nn.Parameter(torch.randn(32, 32, 1, 1))
        self.w2 = torch.nn.Parameter(torch.randn(32, 32, 1, 1))

    def forward(self, x):
        return (_QueuedWgrad.apply(x, self.w1) @ self.w2.flatten(1).t()).sum()


def _deferral_on_cpu(monkeypatch):
    from jtsm_amd.layers import conv

    def wgrad(g, x, w, stride=1, pad=0, dil=1, row_scale=None):       # what the contraction kernel computes
        return (g.t.t() @ x.t).view(w.shape)
    monkeypatch.setattr(conv, "MATH", "bf16x3")
    monkeypatch.setattr(conv, "DEFER_WGRAD", True)
    monkeypatch.setattr(conv, "planes_backward_weight", wgrad)
    return conv


def test_deferred_weight_gradient_counts_once_with_grads_kept_and_two_backwards_per_step(monkeypatch):
    """ADVICE r3 (medium, engine/dp.py): autograd runs a parameter's post-accumulate hook even when the node handed it
    nothing; with zero_grad(set_to_none=False), or in a second backward before the optimizer step, `.grad` is then still
    the OLD gradient — the exchange must not count it (the bucket would go out before the queued gradient exists, the
    counter would go negative).  Every bucket is issued exactly once per backward, behind its real gradient."""
    from jtsm_amd.engine import dp
    conv = _deferral_on_cpu(monkeypatch)

    torch.manual_seed(0)
    model = _TwoLayers()
    ex = dp.GradientExchange(model, torch.device("cpu"), cap_bytes=64, rebucket=False)
    try:
        assert len(ex.buckets) == 2
        x = torch.randn(5, 32)
        ref = _TwoLayers()
        ref.load_state_dict(model.state_dict())
        (ref.w2.flatten(1) @ ref.w1.flatten(1) @ x.t()).sum().backward()     # plain autograd, no deferral
        want = {"w1": ref.w1.grad.clone(), "w2": ref.w2.grad.clone()}
        issued = []
        launch = ex._launch

        def spy(i):       # the bucket's content at the moment its collective would be issued
            issued.append((i, ex.buckets[i].flat.clone()))
            launch(i)
        ex._launch = spy
        # (a) gradients kept across steps (set_to_none=False): .grad is non-None when autograd's empty hook call comes
        for step in range(3):
            model.zero_grad(set_to_none=False)
            del issued[:]
            model(x).backward()
            assert sorted(i for i, _ in issued) == [0, 1] and [i for i, _ in issued] == sorted(i for i, _ in issued)
            assert all(b.pending == len(b.params) for b in ex.buckets) and not conv._DEFERRED
            for name in ("w1", "w2"):
                p = getattr(model, name)
                assert torch.allclose(p.grad, want[name], atol=1e-4), (step, name)
                b, view = ex._slot[p]
                snap = next(s for i, s in issued if ex.buckets[i] is b)
                assert torch.allclose(snap[:p.numel()].view(p.shape), want[name], atol=1e-4), \
                    "bucket of %s issued before its gradient" % name
        # (b) two backward passes into the same gradients (no zero_grad between): the second adds, each pass issues once
        model.zero_grad(set_to_none=False)
        for k in (1, 2):
            del issued[:]
            model(x).backward()
            assert sorted(i for i, _ in issued) == [0, 1]
            assert all(b.pending == len(b.params) for b in ex.buckets)
            assert torch.allclose(model.w1.grad, k * want["w1"], atol=1e-4 * k)
            assert torch.allclose(model.w2.grad, k * want["w2"], atol=1e-4 * k)
    finally:
        ex.detach()


def test_aborted_backward_does_not_leak_queued_weight_gradients_into_the_next_step(monkeypatch):
    """ADVICE r3 (medium, layers/conv.py): a backward that raises never runs the engine's final callbacks; the queue
    must not pin that step's operands, add them into the next step's gradients, or stay 'armed' for ever.  The next
    forward (planes_clear) drops it; the next backward re-arms its own end-of-backward flush."""
    conv = _deferral_on_cpu(monkeypatch)
    torch.manual_seed(1)
    model = _TwoLayers()
    x = torch.randn(5, 32)
    conv.planes_clear()
    dropped = conv.STALE_DROPPED[0]
    _QueuedWgrad.fail = True
    try:
        try:
            model(x).backward()
            raise AssertionError("the backward did not abort")
        except RuntimeError as e:
            assert "aborted" in str(e)
    finally:
        _QueuedWgrad.fail = False
    assert len(conv._DEFERRED) == 1 and conv.deferred_pending(model.w1)
    model.zero_grad(set_to_none=True)
    conv.planes_clear()                                     # what every forward of the model starts with
    assert not conv._DEFERRED and not conv.deferred_pending(model.w1) and conv.STALE_DROPPED[0] == dropped + 1
    x2 = torch.randn(5, 32)
    model(x2).backward()                                    # no stage flush here: only the re-armed callback delivers
    ref = _TwoLayers()
    ref.load_state_dict(model.state_dict())
    (ref.w2.flatten(1) @ ref.w1.flatten(1) @ x2.t()).sum().backward()
    assert model.w1.grad is not None and torch.allclose(model.w1.grad, ref.w1.grad, atol=1e-4)
    assert not conv._DEFERRED
