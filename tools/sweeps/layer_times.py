"""One training step of the bench model with every contraction launch timed: per (role, shape) kernel time, TFLOP/s, GB/s, roofline fraction, and the split-K finishing cost per layer.

Run from the repo root on the GPU box:  python tools/sweeps/layer_times.py
(measurement helper behind the constants quoted in csrc/conv_x3.h / conv_igemm.hip / moi_pool.hip; not part of the product)."""
import sys, os, torch, collections
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import bench
from jtsm_amd.layers import conv
from jtsm_amd.utils.synthetic import synthetic_inputs
dev = torch.device('cuda:0')
model = bench.build(dev)
inputs = synthetic_inputs(1234, batch=2, size=1024, proposals=2000, device=dev)
opt = bench.make_optimizer(model)
def step():
    l = model(inputs); sum(l.values()).backward(); opt.step(); opt.zero_grad(set_to_none=True)
for _ in range(3): step()
torch.cuda.synchronize()
conv.LAUNCH_LOG = []
step(); torch.cuda.synchronize()
log, conv.LAUNCH_LOG = conv.LAUNCH_LOG, None
agg = collections.OrderedDict()
for v, fl, sp, shp in log:
    k = (("W" if "wgrad" in v else v.split("<")[1].split(",")[0]), shp, v)
    d = agg.setdefault(k, [0, 0.0, 0.0]); d[0] += 1; d[1] += fl; d[2] += sp.kernel_ms()
tot = sum(d[2] for d in agg.values())
print("total igemm ms", tot)
for k, d in sorted(agg.items(), key=lambda kv: -kv[1][2])[:90]:
    by = k[1][-1]; 
    print("%-6s %-44s %-38s n=%2d  %7.3f ms  %6.1f TF %6.0f GB/s  roof %.2f (%.1f%%)" % (k[0], k[1][:-1], k[2][9:], d[0], d[2], d[1]/d[2]/1e9, by*d[0]/d[2]/1e6, max(d[1]/833e12, by*d[0]/8e12)*1e3/d[2], 100*d[2]/tot))
print("---- finish (call - kernel) by layer")
fin = collections.OrderedDict()
for v, fl, sp, shp in log:
    k = (("W" if "wgrad" in v else v.split("<")[1].split(",")[0]), shp[:-1], v[9:])
    d = fin.setdefault(k, [0, 0.0]); d[0] += 1; d[1] += max(sp.call_ms() - sp.kernel_ms(), 0.0)
print("total finish ms", sum(d[1] for d in fin.values()))
for k, d in sorted(fin.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-6s %-44s %-38s n=%2d  %7.3f ms" % (k[0], k[1], k[2], d[0], d[1]))
