"""Same-process A/B: the semantic head on a side stream (modeling/meta_arch/mcnn.py: SEM_SIDE_STREAM) against the single
stream with the head inside the mask branch's synchronisation window; alternating blocks of steps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from jtsm_amd.modeling.meta_arch import mcnn
from jtsm_amd.utils.synthetic import synthetic_inputs

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
model = bench.build(dev)
inputs = synthetic_inputs(1234, batch=2, size=1024, proposals=2000, device=dev, cluster=1.0, objects=40)
opt = bench.make_optimizer(model)
last = {}

def step():
    losses = model(inputs)
    total = sum(losses.values())
    total.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
    last.update({k: v.detach() for k, v in losses.items()})      # (no host read-back inside the timed steps)

for _ in range(8):
    step()
res = {0: [], 1: []}
for rnd in range(6):
    for mode in (0, 1):
        mcnn.SEM_SIDE_STREAM = bool(mode)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(25):
            step()
        torch.cuda.synchronize()
        res[mode].append((time.perf_counter() - t0) / 25 * 1e3)
for mode in (0, 1):
    print("semantic head on a side stream = %d: %s ms/step" % (mode, " ".join("%.3f" % x for x in res[mode])))
print("losses", {k: round(float(v), 5) for k, v in sorted(last.items())})
