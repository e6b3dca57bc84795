"""Is the step waiting for the HOST?  A busy-wait of D microseconds is put at one point of the step (before the forward,
between forward and backward, behind the backward, behind the optimizer); if the step grows by D the enqueueing thread is
on the critical path there, if it does not the device had work queued.  Alternating blocks of 20 steps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from jtsm_amd.utils.synthetic import synthetic_inputs

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
model = bench.build(dev)
inputs = synthetic_inputs(1234, batch=2, size=1024, proposals=2000, device=dev, cluster=1.0, objects=40)
opt = bench.make_optimizer(model)
delay = {"where": None, "us": 0.0}


def spin(where):
    if delay["where"] == where and delay["us"] > 0:
        t = time.perf_counter() + delay["us"] * 1e-6
        while time.perf_counter() < t:
            pass


def step():
    spin("before forward")
    losses = model(inputs)
    total = sum(losses.values())
    spin("before backward")
    total.backward()
    spin("behind backward")
    opt.step()
    spin("behind optimizer")
    opt.zero_grad(set_to_none=True)


def measure(n=20):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(8):
    step()
D = float(sys.argv[1]) if len(sys.argv) > 1 else 1000.0
res = {}
for rnd in range(4):
    for where in (None, "before forward", "before backward", "behind backward"):
        delay.update(where=where, us=D if where else 0.0)
        res.setdefault(where, []).append(measure())
for where, v in res.items():
    print("%-18s: %s ms per step (host delay %.0f us there)" % (where or "no delay", " ".join("%.3f" % x for x in v), D if where else 0), flush=True)
