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
