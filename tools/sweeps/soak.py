"""Soak: N training steps of the bench workload; prints step time and device memory every 50 steps (leak check)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from jtsm_amd.layers import conv
from jtsm_amd.utils.synthetic import synthetic_inputs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
model = bench.build(dev)
opt = bench.make_optimizer(model)
batches = [synthetic_inputs(1234 + i, batch=2, size=1024, proposals=2000, device=dev, cluster=1.0, objects=40) for i in range(4)]
t0 = time.perf_counter()
for i in range(n):
    losses = model(batches[i % 4])          # four different batches: the foreground count (a tensor shape) keeps changing
    total = sum(losses.values())
    total.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
    if (i + 1) % 50 == 0:
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("step %4d  %.2f ms/step  alloc %.2f GB  reserved %.2f GB  peak %.2f GB  loss %.4f  plans %d  wplanes %d" % (
            i + 1, 1e3 * dt / 50, torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30,
            torch.cuda.max_memory_allocated() / 2**30, float(total.detach()), len(conv._PLANS), len(conv._WPLANES)), flush=True)
        t0 = time.perf_counter()
