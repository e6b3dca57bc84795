"""Host time of the backward pass by autograd node class (wall time inside each Function.backward on the autograd
thread, N steps), and of the forward by Function.forward: where the ENQUEUE time of the step goes.
usage: host_backward_profile.py [N]"""
import collections, functools, os, sys, time
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
acc = collections.defaultdict(lambda: [0, 0.0])


def wrap(cls, name):
    fn = cls.__dict__[name]
    raw = fn.__func__ if isinstance(fn, staticmethod) else fn

    @functools.wraps(raw)
    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return raw(*a, **k)
        finally:
            e = acc[(name, cls.__name__)]
            e[0] += 1
            e[1] += time.perf_counter() - t0
    setattr(cls, name, staticmethod(timed))


def all_subclasses(c):
    out = []
    for s in c.__subclasses__():
        out.append(s)
        out += all_subclasses(s)
    return out


for cls in all_subclasses(torch.autograd.Function):
    if cls.__module__.startswith("jtsm_amd"):
        for name in ("forward", "backward"):
            if name in cls.__dict__:
                wrap(cls, name)


def step():
    losses = model(inputs)
    sum(losses.values()).backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


for _ in range(6):
    step()
torch.cuda.synchronize()
acc.clear()
t0 = time.perf_counter()
tb = 0.0
for _ in range(N):
    losses = model(inputs)
    total = sum(losses.values())
    t1 = time.perf_counter()
    total.backward()
    tb += time.perf_counter() - t1
    opt.step()
    opt.zero_grad(set_to_none=True)
torch.cuda.synchronize()
print("%.3f ms per step (with the timers); backward() call %.3f ms per step" % ((time.perf_counter() - t0) / N * 1e3, tb / N * 1e3))
for kind in ("backward", "forward"):
    rows = sorted(((v[1], v[0], k[1]) for k, v in acc.items() if k[0] == kind), reverse=True)
    print("%s: %.3f ms per step inside the nodes, %d calls per step" % (kind, sum(r[0] for r in rows) / N * 1e3, sum(r[1] for r in rows) // N))
    for t, n, name in rows[:22]:
        print("   %-34s %5.1f calls/step %8.1f us/call %8.3f ms/step" % (name, n / N, t / n * 1e6, t / N * 1e3))
