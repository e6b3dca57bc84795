"""cProfile of the AUTOGRAD thread (the backward's Python runs there, out of the main thread's profiler): the profiler is
switched on from inside the first backward node that runs on that thread and read out after N steps."""
import cProfile, io, os, pstats, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from jtsm_amd.layers import wsl_losses
from jtsm_amd.utils.synthetic import synthetic_inputs

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
model = bench.build(dev)
inputs = synthetic_inputs(1234, batch=2, size=1024, proposals=2000, device=dev, cluster=1.0, objects=40)
opt = bench.make_optimizer(model)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
prof = {}


class _Probe(torch.autograd.Function):      # the LAST op of the forward = the FIRST node of the backward
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        tid = threading.get_ident()
        if prof.get("on") and tid not in prof:
            prof[tid] = cProfile.Profile()
            prof[tid].enable()
        return g


def step():
    losses = model(inputs)
    total = _Probe.apply(sum(losses.values()))
    total.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


for _ in range(6):
    step()
torch.cuda.synchronize()
prof["on"] = True
for _ in range(N):
    step()
torch.cuda.synchronize()
for tid, p in prof.items():
    if tid == "on":
        continue
    # (the profiler object belongs to the autograd thread; reading its stats from here is fine once the steps are done)
    s = io.StringIO()
    try:
        p.disable()
    except Exception:
        pass
    pstats.Stats(p, stream=s).sort_stats("tottime").print_stats(48)
    txt = s.getvalue().replace(ROOT + "/", "")
    print("\n".join(l[:160] for l in txt.split("\n")[:64]))
