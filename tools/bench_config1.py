#!/usr/bin/env python3
"""BASELINE configs[1] on one MI355X: R50-FPN forward only on 8 x 3 x 1024 x 1024 random tensors, then ROIAlign
(7x7 box pooler and 14x14 mask pooler, sampling_ratio 0, aligned) forward and backward on 512 synthetic proposals
(64 per image, SURVEY §8d recipe), with the HBM roofline of §8d:

    forward bytes  = M*C*P*P*4 written + min(unique feature bytes, sampled bytes) read
    backward bytes = M*C*P*P*4 read + sum_l B*C*H_l*W_l*4 written (the zero-initialised gradient maps)

    python tools/bench_config1.py [--steps 20 --warmup 5]

Prints one JSON object.  Not part of bench.py's contract (that is configs[2]); op-level figures for the README."""
import argparse
import json
import math
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
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--rois-per-image", type=int, default=64)
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("needs a HIP device")
    import bench
    from jtsm_amd.layers.conv import planes_clear
    from jtsm_amd.modeling.poolers import ROIPooler
    from jtsm_amd.structures import Boxes

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    model = bench.build(dev)
    model.eval()
    B, S = args.batch, args.size
    g = torch.Generator().manual_seed(1234)
    images = (torch.rand(B, 3, S, S, generator=g) * 255).to(dev).contiguous(memory_format=torch.channels_last)
    images = images - model.pixel_mean

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    def fwd():
        planes_clear()
        with torch.no_grad():
            return model.backbone(images)

    t_fwd = timed(fwd, args.steps, args.warmup)
    feats = fwd()
    names = ["p2", "p3", "p4", "p5"]
    levels = [feats[n].detach() for n in names]
    gflop = 279.9 * B                                   # 140.0 GMAC per 1024^2 image, forward

    boxes = []
    for _ in range(B):
        n = args.rois_per_image
        x0, y0 = torch.rand(n, generator=g) * S * 0.75, torch.rand(n, generator=g) * S * 0.75
        lo, hi = math.log(16.0), math.log(S / 2.0)
        w = torch.exp(torch.rand(n, generator=g) * (hi - lo) + lo)
        h = torch.exp(torch.rand(n, generator=g) * (hi - lo) + lo)
        boxes.append(Boxes(torch.stack([x0, y0, (x0 + w).clamp(max=S), (y0 + h).clamp(max=S)], 1).to(dev)))
    M, C = B * args.rois_per_image, levels[0].shape[1]
    feat_bytes = sum(f.numel() * 4 for f in levels)
    ops = {}
    for res in (7, 14):
        pooler = ROIPooler(res, [1 / 4, 1 / 8, 1 / 16, 1 / 32], 0, "ROIAlignV2")
        xs = [f.clone().requires_grad_() for f in levels]
        y = pooler(xs, boxes)
        gy = torch.randn_like(y)
        t_f = timed(lambda: pooler(levels, boxes), args.steps, args.warmup)

        def bwd():
            for x in xs:
                x.grad = None
            pooler(xs, boxes).backward(gy)

        t_fb = timed(bwd, args.steps, args.warmup)
        out_bytes = M * C * res * res * 4
        # unique feature bytes: cells of each level under at least one of ITS rois (box grown by one cell for the
        # bilinear neighbours), x C x 4 — §8d's min(unique, sampled) is this for every case here
        from jtsm_amd.modeling.poolers import assign_boxes_to_levels
        lv = assign_boxes_to_levels(boxes, 2, 5, 224, 4).cpu()
        unique_cells, off = 0, 0
        for bi, bx in enumerate(boxes):
            b = bx.tensor.cpu()
            for l, f in enumerate(levels):
                Hl, Wl = f.shape[2], f.shape[3]
                sc = Hl / float(S)
                cover = torch.zeros(Hl, Wl, dtype=torch.bool)
                for r in b[lv[off:off + len(b)] == l]:
                    x0, y0 = max(int(r[0] * sc) - 1, 0), max(int(r[1] * sc) - 1, 0)
                    x1, y1 = min(int(r[2] * sc) + 2, Wl), min(int(r[3] * sc) + 2, Hl)
                    cover[y0:y1, x0:x1] = True
                unique_cells += int(cover.sum())
            off += len(b)
        fb = out_bytes + unique_cells * C * 4
        bb = out_bytes + feat_bytes
        ops["roi_align_%dx%d" % (res, res)] = {
            "forward_us": round(t_f * 1e6, 1), "forward_GB_per_s": round(fb / t_f / 1e9, 1),
            "forward_plus_backward_us": round(t_fb * 1e6, 1),
            "backward_GB_per_s": round(bb / max(t_fb - t_f, 1e-9) / 1e9, 1),
            "algorithmic_bytes": {"forward": fb, "backward": bb}}
    print(json.dumps({
        "config": "BASELINE configs[1]: R50-FPN forward-only, %d x 3x%dx%d, ROIAlign on %d proposals" % (B, S, S, M),
        "backbone_forward": {"images_per_sec": round(B / t_fwd, 2), "ms": round(t_fwd * 1e3, 3),
                             "tflops_fp32_equivalent": round(gflop / t_fwd / 1e3, 1),
                             "frac_of_833_peak": round(gflop / t_fwd / 1e3 / 833.3, 3)},
        "ops": ops, "hbm_peak_GB_per_s": 8000}))


if __name__ == "__main__":
    main()
