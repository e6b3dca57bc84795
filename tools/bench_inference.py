#!/usr/bin/env python3
"""Inference throughput of the JTSM panoptic composite on one MI355X: the same model and
synthetic inputs as bench.py (BASELINE configs[2] shapes), eval mode, full post-processing (K-head averaged
detections, per-class NMS, mask heads, paste, semantic resize + arg-max, panoptic merge).

    python tools/bench_inference.py [--steps 20 --warmup 5] [--cpu-baseline]

Prints one JSON line; `stages_ms` splits one pass with hipEvents on the launch stream.  With --cpu-baseline the
torch-CPU oracle (oracle/inference.py) is timed on one image.  Not part of bench.py's contract (that is the
training metric); used for the README table and profiles/r01_inference_*."""
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
    ap.add_argument("--score-thresh", type=float, default=1e-5)   # projects/WSL/configs/PascalVOC-Detection/oicr_WSR_18_DC5_1x.yaml:24
    ap.add_argument("--nms-thresh", type=float, default=0.3)      # :25
    ap.add_argument("--cpu-baseline", action="store_true")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("needs a HIP device (the product path has no CPU fallback)")
    import bench
    from jtsm_amd.utils.synthetic import synthetic_inputs

    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    model = bench.build(device)
    for r in model.roi_heads.box_refinery:
        r.test_score_thresh, r.test_nms_thresh = args.score_thresh, args.nms_thresh
    model.combine_instances_confidence_threshold = 0.0   # random weights: let every detection reach the merge
    model.eval()
    inputs = synthetic_inputs(1234, batch=args.batch, size=args.size, proposals=args.proposals, device=device)

    def step():
        return model(inputs)

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    # one extra pass split by stage
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    from jtsm_amd.structures import ImageList
    for _ in range(2):   # the second split is reported (the first one warms the allocator for this call pattern)
      with torch.no_grad():
          ev[0].record()
          images = model.preprocess_image(inputs)
          feats = model.backbone(images.tensor)
          ev[1].record()
          sp = ImageList.from_tensors([x["superpixels"].to(device) for x in inputs], model.backbone.size_divisibility)
          props = [x["proposals"].to(device) for x in inputs]
          model.roi_heads.proposals, model.roi_heads.superpixels, model.roi_heads.images = props, sp, images
          dets, all_scores, all_boxes = model.roi_heads._forward_box_inference(feats, props)
          ev[2].record()
          dets, _, _ = model.roi_heads.forward_with_given_boxes(feats, dets)
          ev[3].record()
          sem, _ = model.sem_seg_head(feats, None)
          ev[4].record()
          model._postprocess_ps(sem, dets, inputs, images.image_sizes)
          ev[5].record()
    torch.cuda.synchronize()
    names = ["backbone", "box_branch+detections", "mask_heads", "sem_seg_head", "postprocess(paste,resize,merge)"]
    stages = {n: round(ev[i].elapsed_time(ev[i + 1]), 3) for i, n in enumerate(names)}

    res = {
        "metric": "images/sec inference, R50-FPN JTSM panoptic, 2x1024x1024, 1 GPU", "value": round(args.batch * args.steps / dt, 3),
        "unit": "images/sec", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "data": "synthetic",
        "config": {"workload": "BASELINE configs[2] shapes in eval mode: %d x 3x%dx%d, %d proposals + %d superpixels per "
                               "image; 4-head averaged detections (score > %g, NMS %.1f, top 100), mask head, paste to "
                               "full resolution, 54-class semantic map, panoptic merge" % (
                                   args.batch, args.size, args.size, args.proposals, (args.size // 32) ** 2,
                                   args.score_thresh, args.nms_thresh)},
        "stages_ms": stages,
        "detections_per_image": [len(o["instances"]) for o in out],
        "segments_per_image": [len(o["panoptic_seg"][1]) for o in out],
    }
    if args.cpu_baseline:
        from oracle import inference as OI
        from oracle import model as OM
        p = OM.init_params(0, input_gain=1.0 / 64)
        b = OM.synthetic_batch(1234, B=1, size=args.size, R=args.proposals, sp_block=32)
        t0 = time.perf_counter()
        OI.forward_inference(p, b, score_thresh=args.score_thresh, nms_thresh=args.nms_thresh,
                             instances_confidence_threshold=0.0)
        cdt = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": round(1.0 / cdt, 4), "unit": "images/sec", "cores": torch.get_num_threads(),
                               "kind": "port", "sample": "oracle/inference.py forward_inference on 1 image (%.1f s)" % cdt}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
