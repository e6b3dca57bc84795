#!/usr/bin/env python3
"""Host -> HBM leg of the input contract under the training step of bench.py: the same model and
step, but every step consumes a FRESH host batch (uint8 image, proposals, superpixels, labels) delivered by
jtsm_amd.data.DevicePrefetcher (one pinned arena, one async copy per batch on a side stream, double buffered), the
normalise + pad + channels-last conversion done by one launch on the device.

    python tools/bench_input_pipeline.py [--steps 20 --warmup 5]

Prints one JSON line: images/sec with the link in the loop (the PCIe-inclusive rate DESIGN.md quotes next to
bench.py's resident-input value), bytes per batch on the link and the bare copy rate."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--proposals", type=int, default=2000)
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("needs a HIP device")
    import bench
    from jtsm_amd.data import DevicePrefetcher
    from jtsm_amd.utils.synthetic import synthetic_inputs

    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    model = bench.build(device)
    opt = bench.make_optimizer(model)
    n = args.steps + args.warmup
    pool = []
    for seed in range(4):   # four distinct host batches, cycled (the link does not care about content)
        b = synthetic_inputs(1234 + seed, batch=args.batch, size=args.size, proposals=args.proposals, device="cpu")
        for x in b:
            x["image"] = x["image"].clamp(0, 255).to(torch.uint8)     # what DatasetMapper emits
        pool.append(b)
    batches = [pool[i % len(pool)] for i in range(n)]
    pre = DevicePrefetcher(batches, device, depth=2)

    def step(inputs):
        losses = model(inputs)
        sum(losses.values()).backward()
        opt.step()
        opt.zero_grad(set_to_none=True)

    t0 = None
    for i, inputs in enumerate(pre):
        if i == args.warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        step(inputs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    # bare link rate of one batch arena
    slot = pre.slots[0]
    nbytes = pre.bytes_last
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(10):
        slot.device[:nbytes].copy_(slot.pinned[:nbytes], non_blocking=True)
    ev1.record()
    torch.cuda.synchronize()
    copy_ms = ev0.elapsed_time(ev1) / 10

    print(json.dumps({
        "metric": "images/sec training with the host->HBM link in the loop, R50-FPN JTSM panoptic, 2x1024x1024, 1 GPU",
        "value": round(args.batch * args.steps / dt, 3), "unit": "images/sec", "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 3),
        "link": {"bytes_per_batch": int(nbytes), "copy_ms": round(copy_ms, 3),
                 "GB_per_s": round(nbytes / copy_ms / 1e6, 2),
                 "note": "uint8 image + uint8-narrowed oh_labels / sem_seg + int32 superpixels + boxes in one pinned "
                         "arena; the copy runs on a side stream under the previous step"}}))


if __name__ == "__main__":
    main()
