"""Which tensors still get a standalone bf16 split / gated-gradient pass in one training step (plane emission coverage).

Run from the repo root on the GPU box:  python tools/sweeps/split_log.py
(measurement helper behind the constants quoted in csrc/conv_x3.h / conv_igemm.hip / moi_pool.hip; not part of the product)."""
import sys, os, torch, collections, traceback
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import bench
from jtsm_amd.layers import conv, elementwise
from jtsm_amd.utils.synthetic import synthetic_inputs
dev = torch.device('cuda:0')
model = bench.build(dev)
inputs = synthetic_inputs(1234, batch=2, size=1024, proposals=2000, device=dev, cluster=1.0, objects=40)
opt = bench.make_optimizer(model)
def step():
    l = model(inputs); sum(l.values()).backward(); opt.step(); opt.zero_grad(set_to_none=True)
for _ in range(2): step()
log = collections.Counter()
orig = conv._split
def spy(t, *a):
    fr = [f for f in traceback.extract_stack()[:-1] if 'jtsm_amd' in f.filename][-3:]
    log[(tuple(t.shape), " < ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(fr)))] += 1
    return orig(t, *a)
conv._split = spy
orig_r = elementwise.relu_backward
rlog = collections.Counter()
def spy_r(dy, y, emit_planes=False):
    fr = [f for f in traceback.extract_stack()[:-1] if 'jtsm_amd' in f.filename][-2:]
    rlog[(tuple(dy.shape), emit_planes, " < ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(fr)))] += 1
    return orig_r(dy, y, emit_planes)
elementwise.relu_backward = spy_r
import jtsm_amd.layers.conv as C2
if hasattr(C2, 'relu_backward'): C2.relu_backward = spy_r
import jtsm_amd.layers.fused_blocks as FB
FB.relu_backward = spy_r
step(); torch.cuda.synchronize()
print("---- standalone splits")
for (shape, where), n in sorted(log.items(), key=lambda kv: -torch.Size(kv[0][0]).numel() * kv[1]):
    print(n, shape, "%.1f MB" % (torch.Size(shape).numel() * 4 / 1e6), where)
print("---- relu_backward")
for (shape, ep, where), n in sorted(rlog.items(), key=lambda kv: -torch.Size(kv[0][0]).numel() * kv[1]):
    print(n, shape, ep, "%.1f MB" % (torch.Size(shape).numel() * 4 / 1e6), where)
