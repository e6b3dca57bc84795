"""Where the HOST spends a training step (cProfile over N steps of the benchmarked workload): the step is enqueue-bound
in places (tools/sweeps/gpu_idle.py shows the device waiting at the step boundary and behind the backward's end).
usage: host_profile.py [N] [sort: tottime|cumtime]"""
import cProfile, io, os, pstats, sys, time
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def step():
    losses = model(inputs)
    sum(losses.values()).backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


for _ in range(6):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    step()
torch.cuda.synchronize()
print("untraced: %.3f ms per step" % ((time.perf_counter() - t0) / N * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    step()
torch.cuda.synchronize()
pr.disable()
for key in (sys.argv[2:] or ["tottime", "cumtime"]):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    txt = s.getvalue().replace(ROOT + "/", "")
    print("\n".join(l[:170] for l in txt.split("\n")[:70]))
