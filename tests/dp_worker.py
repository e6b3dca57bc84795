"""Worker of tests/test_hip_dp.py: one rank of a data-parallel training step of the REAL model.

    python tests/dp_worker.py RANK WORLD PORT OUTDIR [collective]

WORLD > 1: ranks share GPU 0 and talk over gloo (RCCL refuses two ranks on one device; the exchange code path —
buckets, kernel-side gradient slots, hooks, side stream, end-of-backward callback — is the one bench.py runs).
WORLD == 0 (the reference computation): a single process runs BOTH shards, averages the two gradients itself and takes
the same optimizer step.  WORLD == 1: ONE rank over the real RCCL backend ("nccl") with the collectives forced — on a
single device this is how reduce_scatter_tensor(AVG) / all_gather_into_tensor, the side stream and the event ordering
of the exchange are driven through RCCL itself; WORLD == -1 is its reference (the same shard, no exchange).  Saves the parameters before (`init`) and after (`params`) the step to OUTDIR/rank{RANK}.pt."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build(device):
    from model_util import jtsm_cfg
    from jtsm_amd.modeling import build_model

    torch.manual_seed(0)
    model = build_model(jtsm_cfg(str(device)))
    model.train()
    model.roi_heads.box_head.dropout_p = 0.0
    with torch.no_grad():
        model.backbone.bottom_up.stem.conv1.weight.mul_(1.0 / 64)
    return model


def optimizer(model):
    from jtsm_amd.solver import SGD

    decay = [p for n, p in model.named_parameters() if p.requires_grad and not n.endswith(".bias")]
    bias = [p for n, p in model.named_parameters() if p.requires_grad and n.endswith(".bias")]
    return SGD([{"params": decay, "lr": 1e-3, "weight_decay": 5e-4}, {"params": bias, "lr": 2e-3, "weight_decay": 0.0}],
               lr=1e-3, momentum=0.9)


def shard(rank, device):
    from jtsm_amd.utils.synthetic import synthetic_inputs
    return synthetic_inputs(1234 + rank, batch=2, size=256, proposals=120, sp_block=8, device=device, cluster=0.2)


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    if os.environ.get("DPW_FORGET_PRODUCERS"):     # ad-hoc negative control: the exchange no longer learns of the side streams
        from jtsm_amd.layers import conv
        conv.register_producer_stream = lambda stream: None
    collective = sys.argv[5] if len(sys.argv) > 5 else None
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    model = build(device)
    opt = optimizer(model)
    init = {n: p.detach().cpu().clone() for n, p in model.named_parameters() if p.requires_grad}
    if world == 1:
        import torch.distributed as dist
        from jtsm_amd.engine import dp
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
        ex = dp.GradientExchange(model, device, collective or "rs_ag", force_collectives=True)
        assert ex.comm_stream is not None and ex.collective == (collective or "rs_ag")
        for _ in range(2):     # twice: the end-of-backward callback re-arms the buckets
            model.zero_grad(set_to_none=True)
            losses = model(shard(0, device))
            sum(losses.values()).backward()
        for p in ex._slot:
            assert p.grad is not None and p.grad.data_ptr() == ex._slot[p][1].data_ptr()
        info = {"buckets": [b.numel for b in ex.buckets], "bytes": ex.bytes}
    elif world == -1:
        losses = model(shard(0, device))
        sum(losses.values()).backward()
        info = {}
    elif world > 1:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        from jtsm_amd.engine import dp
        dp.init_distributed("gloo", device)
        net = dp.wrap_data_parallel(model, device, collective)
        assert isinstance(net, dp.DataParallel)
        from jtsm_amd.layers import conv as _conv
        _orig, _count = _conv.side_weight_gradients, [0, 0]

        def _counting(params, compute, operands=()):
            ok = _orig(params, compute, operands)
            _count[0 if ok else 1] += 1
            return ok
        _conv.side_weight_gradients = _counting
        losses = net(shard(rank, device))
        if os.environ.get("DPW_DELAY_SIDE_STREAMS"):
            # every side stream that produces gradients starts the backward ~60 ms late (a spin kernel in front of its
            # work): a collective that did not wait for it would average a bucket the stream has not written yet
            from jtsm_amd.layers import conv
            conv._wgrad_side_stream(device)
            for st in list(conv.PRODUCER_STREAMS) or [conv._wgrad_side_stream(device)]:
                with torch.cuda.stream(st):
                    torch.cuda._sleep(int(1.5e8))
        sum(losses.values()).backward()
        ex = net.exchange
        # every trainable parameter's gradient now lives in its flat bucket, with the parameter's own strides
        for p in ex._slot:
            assert p.grad is not None and p.grad.data_ptr() == ex._slot[p][1].data_ptr() and p.grad.stride() == p.stride()
        info = {"buckets": [b.numel for b in ex.buckets], "bytes": ex.bytes, "loss": float(sum(losses.values()).detach()),
                "rebucketed": ex.rebucketed, "order": [ex._names[p] for b in ex.buckets for p in b.params],
                "side_weight_gradients": {"on_side_stream": _count[0], "declined": _count[1],
                                          "producer_streams": len(_conv.PRODUCER_STREAMS)}}
        print("rank %d side_weight_gradients %s" % (rank, info["side_weight_gradients"]), flush=True)
    else:
        grads = []
        for r in range(2):
            model.zero_grad(set_to_none=True)
            losses = model(shard(r, device))
            sum(losses.values()).backward()
            grads.append({n: p.grad.detach().clone() for n, p in model.named_parameters() if p.requires_grad})
        for n, p in model.named_parameters():
            if p.requires_grad:
                p.grad = torch.empty_like(p).copy_((grads[0][n] + grads[1][n]) / 2)
        info = {}
    opt.step()
    torch.cuda.synchronize()
    state = {n: p.detach().cpu() for n, p in model.named_parameters() if p.requires_grad}
    torch.save({"params": state, "init": init, "info": info}, os.path.join(out, "rank%d.pt" % rank))
    if world > 1:
        torch.distributed.barrier()
    if world >= 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
