"""Same-process A/B: the queued (grouped) weight gradients on a side stream beside the backward (layers/conv.py:
WGRAD_STREAM) against the compute stream; alternating blocks of steps, ms per step per setting; losses must agree."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from jtsm_amd.layers import conv as K
from jtsm_amd.utils.synthetic import synthetic_inputs

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
model = bench.build(dev)
inputs = synthetic_inputs(1234, batch=2, size=1024, proposals=2000, device=dev, cluster=1.0, objects=40)
opt = bench.make_optimizer(model)
last = [None]

def step():
    losses = model(inputs)
    total = sum(losses.values())
    total.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
    last[0] = total

for _ in range(8):
    step()
res = {0: [], 1: []}
for rnd in range(4):
    for mode in (0, 1):
        K.WGRAD_STREAM = bool(mode)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(25):
            step()
        torch.cuda.synchronize()
        res[mode].append((time.perf_counter() - t0) / 25 * 1e3)
for mode in (0, 1):
    print("weight gradients on the side stream = %d: %s ms/step" % (mode, " ".join("%.3f" % x for x in res[mode])))
print("final loss", float(last[0]))
