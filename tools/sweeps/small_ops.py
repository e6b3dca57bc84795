"""Which Python lines launch the small torch device ops of a training step (aten::copy_, fill_, cat, add, ...)."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from torch.profiler import ProfilerActivity, profile
import bench
from jtsm_amd.utils.synthetic import synthetic_inputs

dev = torch.device("cuda", 0)
model = bench.build(dev)
opt = bench.make_optimizer(model)
inputs = synthetic_inputs(1234, batch=2, size=1024, proposals=2000, device=dev, cluster=1.0, objects=40)


def step():
    losses = model(inputs)
    total = sum(losses.values())
    total.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type.name != "CPU" or not ev.name.startswith("aten::"):
        continue
    dur = getattr(ev, "device_time_total", None)
    if dur is None:
        dur = getattr(ev, "cuda_time_total", 0)
    if dur <= 0 or ev.cpu_children and any(c.name.startswith("aten::") and getattr(c, "device_time_total", getattr(c, "cuda_time_total", 0)) > 0 for c in ev.cpu_children):
        continue
    where = "?"
    for fr in (ev.stack or []):
        if "jtsm_amd/" in fr or "bench.py" in fr or "tests/" in fr or "small_ops.py" in fr:
            where = fr[fr.find("jtsm_amd/"):] if "jtsm_amd/" in fr else fr[-80:]
            break
    if where == "?" and ev.stack:
        where = "| ".join(f[-50:] for f in ev.stack[:2])
    k = (ev.name, where)
    agg[k][0] += 1
    agg[k][1] += dur
tot = sum(v[1] for v in agg.values())
print("total device us of leaf aten ops: %.0f" % tot)
for (name, where), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%7.1f us  n=%3d  %-28s %s" % (us, n, name, where[:110]))
