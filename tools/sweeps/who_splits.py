"""Which tensors get a stand-alone plane split (layers/conv.py: _split) in one training step, and from where."""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from jtsm_amd.layers import conv as K
from jtsm_amd.utils.synthetic import synthetic_inputs

dev = torch.device("cuda", 0)
model = bench.build(dev)
inputs = synthetic_inputs(1234, batch=2, size=1024, proposals=2000, device=dev, cluster=1.0, objects=40)
opt = bench.make_optimizer(model)


def step():
    losses = model(inputs)
    sum(losses.values()).backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


for _ in range(4):
    step()
torch.cuda.synchronize()
seen = collections.Counter()
orig = K._split


def spy(t, grad=False, *a, **k):
    st = [f for f in traceback.extract_stack()[:-1] if "jtsm_amd" in f.filename][-4:]
    seen[(tuple(t.shape), bool(grad), " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(st)))] += 1
    return orig(t, grad, *a, **k)


K._split = spy
step()
torch.cuda.synchronize()
K._split = orig
for (shape, grad, where), n in sorted(seen.items(), key=lambda kv: -kv[1] * torch.Size(kv[0][0]).numel()):
    print("%d x %-26s grad=%d  %.1f MB   %s" % (n, shape, grad, torch.Size(shape).numel() * 4 / 1e6, where))
