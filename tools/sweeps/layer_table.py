"""Per-launch table of one instrumented training step: kernel instantiation, role, shape, time, roofs.
    python tools/sweeps/layer_table.py [--min-us 20]"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--min-us", type=float, default=15.0)
    a = ap.parse_args()
    from jtsm_amd import _lib
    from jtsm_amd.layers import conv
    from jtsm_amd.utils.synthetic import synthetic_inputs
    device = torch.device("cuda", 0)
    model = bench.build(device)
    inputs = synthetic_inputs(1234, batch=2, size=1024, proposals=2000, device=device, cluster=1.0, objects=40)
    opt = bench.make_optimizer(model)

    def step():
        losses = model(inputs)
        sum(losses.values()).backward()
        opt.step()
        opt.zero_grad(set_to_none=True)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    conv.LAUNCH_LOG, _lib.TIMING = [], []
    step()
    torch.cuda.synchronize()
    log, conv.LAUNCH_LOG = conv.LAUNCH_LOG, None
    _lib.TIMING = None
    agg = collections.OrderedDict()
    for variant, flops, span, shape, finish_bytes in log:
        key = (str(variant), getattr(variant, "splits", 1)) + tuple(shape[:-1])
        d = agg.setdefault(key, [0, 0.0, 0.0, flops, shape[-1], finish_bytes])
        d[0] += 1
        d[1] += span.kernel_ms()
        d[2] += max(span.call_ms() - span.kernel_ms(), 0.0)
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    print("%-42s %3s %-34s %3s %8s %8s %7s %7s %6s" % ("kernel", "spl", "shape (N,H,W,Cin,Cout,k,s)", "n", "us/launch",
                                                    "fin us", "TF/s", "GB/s", "roof"))
    for key, (n, ms, fin, flops, nbytes, fb) in rows:
        us = 1e3 * ms / n
        if us < a.min_us:
            continue
        peak = bench._mfma_peak(key[0])
        roof = max(flops / (peak * 1e12), nbytes / 8e12) * 1e6
        print("%-42s %3d %-34s %3d %8.1f %8.1f %7.1f %7.0f %6.2f" % (key[0], key[1], str(key[2:]), n, us, 1e3 * fin / n,
                                                                   flops / us / 1e6, nbytes / us / 1e3, roof / us))


if __name__ == "__main__":
    main()
