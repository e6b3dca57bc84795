"""Same-process A/B: the heads' share of the optimizer step early, on the side stream (solver/build.py: attach_early_heads),
against the whole step at the end; alternating blocks of 25 steps."""
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
ids = opt._early_ids


def step():
    losses = model(inputs)
    sum(losses.values()).backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


for _ in range(8):
    step()
res = {0: [], 1: []}
for rnd in range(6):
    for mode in (0, 1):
        opt._early_ids = ids if mode else set()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(25):
            step()
        torch.cuda.synchronize()
        res[mode].append((time.perf_counter() - t0) / 25 * 1e3)
for mode in (0, 1):
    print("heads' update early = %d: %s ms/step" % (mode, " ".join("%.3f" % x for x in res[mode])))
