"""The gradient exchange with the side streams ON: K optimizer steps from the same weights, without an exchange and
under a one-rank RCCL group with the collectives forced (reduce-scatter + all-gather are then copies, issued on the
communication stream behind events of every producer stream) — the trajectories must be bit-identical, and stay so
over repeats: a collective that ran before a side stream had written its bucket would show up as a difference.
usage: exchange_soak.py [K] [REPEATS] [rs_ag|allreduce]"""
import copy, os, socket, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.distributed as dist
import bench
from jtsm_amd.engine import dp
from jtsm_amd.layers import conv as K
from jtsm_amd.modeling.meta_arch import mcnn
from jtsm_amd.utils.synthetic import synthetic_inputs

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
model = bench.build(dev)
inputs = synthetic_inputs(1234, batch=2, size=1024, proposals=2000, device=dev, cluster=1.0, objects=40)
init = copy.deepcopy(model.state_dict())
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
REPEATS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
COLL = sys.argv[3] if len(sys.argv) > 3 else "rs_ag"


def trajectory():
    model.load_state_dict(init)
    opt = bench.make_optimizer(model)
    torch.manual_seed(7)
    out = []
    for _ in range(STEPS):
        losses = model(inputs)
        sum(losses.values()).backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        out.append(torch.stack([losses[k].detach() for k in sorted(losses)]))
    torch.cuda.synchronize()
    return torch.stack(out), torch.cat([p.detach().flatten()[:4096] for p in model.parameters() if p.requires_grad])


print("side streams: weight gradients %s, semantic head %s" % (K.WGRAD_STREAM, mcnn.SEM_SIDE_STREAM), flush=True)
ref, ref_w = trajectory()
again, again_w = trajectory()
print("no exchange, twice: identical", bool(torch.equal(ref, again) and torch.equal(ref_w, again_w)), flush=True)
ex = dp.GradientExchange(model, dev, COLL, force_collectives=True)
if os.environ.get("EXSOAK_FORGET_PRODUCERS"):     # negative control: the collectives no longer wait for the side streams
    del K.PRODUCER_STREAMS[:]
assert ex.comm_stream is not None
bad = 0
for r in range(REPEATS):
    t, w = trajectory()
    same = bool(torch.equal(t, ref) and torch.equal(w, ref_w))
    bad += not same
    side_used = [s for s in K.PRODUCER_STREAMS]
    print("exchange (%s, %d buckets, %d producer streams) repeat %d: identical to the run without it: %s%s" % (
        COLL, len(ex.buckets), len(side_used), r, same,
        "" if same else " (first differing step %d, largest loss difference %.3e)" % (
            int((t != ref).any(dim=1).nonzero()[0]) if not torch.equal(t, ref) else -1, float((t - ref).abs().max()))), flush=True)
for p in ex._slot:
    assert p.grad is None or p.grad.data_ptr() == ex._slot[p][1].data_ptr()
# step time under the (one-rank, forced) exchange with the side streams on and off, alternating
import time
opt = bench.make_optimizer(model)


def steps(n):
    for _ in range(n):
        losses = model(inputs)
        sum(losses.values()).backward()
        opt.step()
        opt.zero_grad(set_to_none=True)


res = {True: [], False: []}
for rnd in range(3):
    for on in (True, False):
        K.WGRAD_STREAM = mcnn.SEM_SIDE_STREAM = on
        steps(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps(20)
        torch.cuda.synchronize()
        res[on].append((time.perf_counter() - t0) / 20 * 1e3)
for on in (True, False):
    print("under the exchange, side streams %s: %s ms/step" % ("on" if on else "off", " ".join("%.3f" % x for x in res[on])), flush=True)
print("exchange soak: %d of %d trajectories differ" % (bad, REPEATS))
dist.destroy_process_group()
