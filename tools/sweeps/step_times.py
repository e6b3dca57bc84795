"""Per-step wall time of the bench step in a FRESH process (how many steps the device takes to reach its steady
clock): python tools/sweeps/step_times.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from jtsm_amd.utils.synthetic import synthetic_inputs  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
device = torch.device("cuda", 0)
model = bench.build(device)
inputs = synthetic_inputs(1234, batch=2, size=1024, proposals=2000, device=device, cluster=1.0, objects=40)
opt = bench.make_optimizer(model)
times = []
for i in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = model(inputs)
    sum(losses.values()).backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    times.append(1e3 * (time.perf_counter() - t0))
print("ms per step (each synchronised):", " ".join("%.1f" % t for t in times))
