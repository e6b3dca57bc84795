#!/usr/bin/env python3
"""bench.py — images/sec of the JTSM hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full training step (forward, backward, gradient all-reduce, SGD update) of the R50-FPN
JTSM panoptic composite on one synthetic batch per rank: BASELINE.json configs[2]
(2 x 3x1024x1024 images, 2000 proposals and 1024 superpixels per image).  Inputs are
resident in HBM before the timed region.  Rank 0 prints ONE JSON line; `value` is the whole-job
images/sec (max step time over ranks).  Extra objects: `roofline` (dominant kernel, measured live with
events on the launch stream in one extra, untimed step) and, at N=1, `cpu_baseline` (the CPU oracle
timed on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same guide: "Peak BF16/FP16 MFMA ~2.5 PF dense"
HBM_PEAK_GBS = 8000.0          # same guide: HBM3E ~8 TB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=1024, help="image side (BASELINE: 1024)")
    ap.add_argument("--proposals", type=int, default=2000, help="proposals per image (BASELINE: 2000)")
    ap.add_argument("--batch", type=int, default=2, help="images per GPU (BASELINE: 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-exact", action="store_true", help="skip the exact-fp32 reference leg")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank code-path rehearsal on a single GPU: every rank uses device 0 and the gloo backend "
                         "(not a measurement)")
    return ap.parse_args()


def build(device):
    from model_util import jtsm_cfg
    from jtsm_amd.modeling import build_model

    torch.manual_seed(0)                       # identical random-init weights on every rank
    model = build_model(jtsm_cfg(str(device)))
    model.train()
    # random msra weights are not matched to 0..255 inputs; keep activations O(1) (as in the parity tests)
    with torch.no_grad():
        model.backbone.bottom_up.stem.conv1.weight.mul_(1.0 / 64)
    return model


def make_optimizer(model, fused=True):
    """SGD as the reference configures it (detectron2/solver/build.py:110-195 with the JTSM config's
    BASE_LR 0.01, WEIGHT_DECAY 5e-4, BIAS_LR_FACTOR 2, WEIGHT_DECAY_BIAS 0, momentum 0.9)."""
    decay, bias = [], []
    for n, p in model.named_parameters():
        if p.requires_grad:
            (bias if n.endswith(".bias") else decay).append(p)
    # Random-init weights diverge within a few steps at the reference's BASE_LR (0.01), which would empty
    # the foreground set and silently shrink the work; the update arithmetic is kept, only lr is tiny.
    lr = 1e-7
    groups = [{"params": decay, "lr": lr, "weight_decay": 5e-4}, {"params": bias, "lr": 2 * lr, "weight_decay": 0.0}]
    if fused:   # the same update for every parameter in one launch (jtsm_amd/solver/build.py)
        from jtsm_amd.solver import SGD
        return SGD(groups, lr=lr, momentum=0.9)
    return torch.optim.SGD(groups, lr=lr, momentum=0.9)


def roofline_leg(step_fn):
    from jtsm_amd.layers import conv

    conv.LAUNCH_LOG = []
    step_fn()
    torch.cuda.synchronize()
    log, conv.LAUNCH_LOG = conv.LAUNCH_LOG, None
    per = {}
    finish_ms = 0.0
    for variant, flops, span, shape in log:
        d = per.setdefault(variant, {"launches": 0, "flops": 0.0, "bytes": 0.0, "ms": 0.0, "roof_ms": 0.0,
                                     "hbm_bound_ms": 0.0})
        # split-bf16 kernels issue three bf16 MFMA products per algorithmic (fp32) product: price the algorithmic
        # rate against a third of the dense bf16 peak
        mfma_peak = BF16_MFMA_PEAK_TFLOPS / 3.0 if "_x3_" in variant else FP32_MFMA_PEAK_TFLOPS
        nbytes = float(shape[-1]) if shape is not None else 0.0
        k = span.kernel_ms()            # the contraction kernel alone (library hook), as rocprofv3 reports it
        t_mfma = flops / (mfma_peak * 1e12) * 1e3
        t_hbm = nbytes / (HBM_PEAK_GBS * 1e9) * 1e3
        d["launches"] += 1
        d["flops"] += flops
        d["bytes"] += nbytes
        d["ms"] += k
        d["roof_ms"] += max(t_mfma, t_hbm)          # the tighter roof of THIS launch's shape
        if t_hbm > t_mfma:
            d["hbm_bound_ms"] += k
        finish_ms += max(span.call_ms() - k, 0.0)   # its split-K finishing pass, when there is one
    for d in per.values():
        d["tflops"] = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
        d["gbs"] = d["bytes"] / (d["ms"] * 1e-3) / 1e9 if d["ms"] > 0 else 0.0
        d["avg_us"] = 1e3 * d["ms"] / d["launches"]
    dom = max(per, key=lambda k: per[k]["ms"])
    D = per[dom]
    x3 = "_x3_" in dom
    peak = BF16_MFMA_PEAK_TFLOPS / 3.0 if x3 else FP32_MFMA_PEAK_TFLOPS
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.isfile(pmc):   # HBM bytes per launch from rocprofv3 --pmc passes of this same command
        t = json.load(open(pmc)).get("kernels", {}).get(dom)
        if t:
            traffic = {"hbm_bytes_per_launch": t["hbm_bytes_per_launch"], "source": "profiles/pmc_traffic.json",
                       "note": t.get("note", "")}
    total_ms = sum(d["ms"] for d in per.values())
    total_fl = sum(d["flops"] for d in per.values())
    # One instantiation serves shapes on both sides of the ridge (res2's 64-channel layers are HBM-bound, res4's
    # are MFMA-bound): the kernel's `bound` is the roof that holds for the larger share of its time.
    hbm = D["hbm_bound_ms"] > 0.5 * D["ms"]
    out = {"bound": "hbm", "kernel": dom, "achieved": round(D["gbs"], 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(D["gbs"] / HBM_PEAK_GBS, 4)} if hbm else \
          {"bound": "mfma", "kernel": dom, "achieved": round(D["tflops"], 2), "peak": round(peak, 1),
           "unit": "TFLOP/s", "frac": round(D["tflops"] / peak, 4)}
    out.update({
        "traffic": traffic,
        "peak_note": ("mfma roof: dense bf16 MFMA peak %.0f TFLOP/s / 3 bf16 products per fp32 product "
                      "(csrc/conv_x3.h), algorithmic fp32 FLOPs / time" % BF16_MFMA_PEAK_TFLOPS if x3 else
                      "mfma roof: dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32)") +
                     "; hbm roof: 8 TB/s, algorithmic bytes = each operand read once + the result written once",
        "mfma_frac": round(D["tflops"] / peak, 4), "hbm_frac": round(D["gbs"] / HBM_PEAK_GBS, 4),
        "hbm_bound_share_of_time": round(D["hbm_bound_ms"] / D["ms"], 3) if D["ms"] > 0 else None,
        # per-launch roofline: sum over launches of max(flops/peak_mfma, bytes/peak_hbm) / measured time
        "roofline_time_frac": round(D["roof_ms"] / D["ms"], 4) if D["ms"] > 0 else None,
        "launches_per_step": D["launches"], "avg_launch_us": round(D["avg_us"], 2),
        "algorithmic_gflop_per_launch": round(D["flops"] / D["launches"] / 1e9, 3),
        "algorithmic_mb_per_launch": round(D["bytes"] / D["launches"] / 1e6, 2),
        "all_contractions": {"tflops": round(total_fl / ((total_ms + finish_ms) * 1e-3) / 1e12, 2),
                             "ms_per_step": round(total_ms + finish_ms, 3), "splitk_finish_ms": round(finish_ms, 3),
                             "gflop_per_step": round(total_fl / 1e9, 1),
                             "roofline_time_frac": round(sum(v["roof_ms"] for v in per.values()) / total_ms, 4),
                             "by_kernel": {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                                               "tflops": round(v["tflops"], 2), "gbs": round(v["gbs"], 1),
                                               "roofline_time_frac": round(v["roof_ms"] / v["ms"], 4)}
                                           for k, v in sorted(per.items()) if v["ms"] > 0}},
    })
    return out


def cpu_baseline_leg(size, proposals):
    """The CPU oracle (oracle/model.py, a torch-CPU port of the reference's arithmetic) on a bounded
    sample: ONE image of the same workload, forward + backward, all host threads."""
    from oracle import model as OM

    p = OM.init_params(0, input_gain=1.0 / 64)
    for n in OM.trainable_names(p):
        p[n].requires_grad_(True)
    b = OM.synthetic_batch(1234, B=1, size=size, R=proposals, sp_block=32)
    t0 = time.perf_counter()
    losses = OM.forward_losses(p, b)
    sum(losses.values()).backward()
    dt = time.perf_counter() - t0
    return {"value": round(1.0 / dt, 4), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "1 training step (fwd+bwd, no optimizer) of the torch-CPU oracle on 1 image %dx%d with %d "
                      "proposals; pooling ops single-threaded C like the reference (%.1f s)" % (size, size, proposals, dt)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product path has no CPU fallback)")
    if args.rehearse_on_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    from jtsm_amd.engine import dp
    dp.init_distributed("gloo" if args.rehearse_on_one_gpu else "nccl", device)
    assert world == args.gpus, "WORLD_SIZE (%d) != --gpus (%d): launch with torch.distributed.run" % (world, args.gpus)

    from jtsm_amd.utils.synthetic import synthetic_inputs

    from jtsm_amd.layers import conv as conv_layers
    conv_math = conv_layers.MATH
    model = build(device)
    inputs = synthetic_inputs(1234 + rank, batch=args.batch, size=args.size, proposals=args.proposals, device=device)
    # one process per GPU; bucketed RCCL all-reduce of gradients overlapped with the backward
    net = dp.wrap_data_parallel(model, device)
    opt = make_optimizer(model)

    def step():
        losses = net(inputs)
        total = sum(losses.values())
        total.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        return total

    def fence():
        dp.fence(device)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    fence()
    dt = time.perf_counter() - t0
    dt = dp.max_over_ranks(dt, device)
    loss_value = float(last.detach())

    out = None
    if rank == 0:
        ims = args.batch * world * args.steps / dt
        out = {
            "metric": "images/sec training, R50-FPN JTSM panoptic, 2x1024x1024, 1/2/4/8 GPU",
            "value": round(ims, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": conv_math, "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[2]: projects/WSL JTSM panoptic R50-FPN composite, COCO-shaped synthetic, "
                            "%d x 3x%dx%d per GPU, %d proposals + %d superpixels per image; MIL + 4 OICR refinements + "
                            "2 mask heads + sem-seg head; fwd + bwd + all-reduce + SGD" % (
                                args.batch, args.size, args.size, args.proposals, (args.size // 32) ** 2),
                "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                "substitutions": "grabCut/polygon pseudo-masks -> eroded pseudo-GT rectangles (SURVEY F8, §8d); dropout on",
                "weights": "random init (msra/xavier as the reference), FrozenBN identity, stem x1/64",
                "math": ("contractions in split-bf16: fp32 operands -> bf16 hi+lo planes, a_lo*b_hi + a_hi*b_lo + a_hi*b_hi "
                         "on v_mfma_f32_32x32x16_bf16 with fp32 accumulate; measured error <= 6e-6 relative per layer "
                         "against fp64 (bar 1e-4); everything else fp32. JTSM_CONV_MATH=f32 selects exact fp32 MFMA "
                         "(see `exact_fp32`)") if conv_math != "f32" else "exact fp32 MFMA contractions",
                "torch_device_ops": ["dropout", "DDP all-reduce", "sort / gather / rasterisation glue of the label path"],
                "final_loss": round(loss_value, 5), "lr": 1e-7,
                "foreground_rois_last_step": int(model.roi_heads.aux["fg_classes"].numel()),
            },
        }
    if not args.no_roofline:
        # every rank runs the extra (untimed) step — it contains the gradient all-reduce — rank 0 reports it
        roof = roofline_leg(step)
        if rank == 0:
            out["roofline"] = roof
    if world == 1 and conv_math != "f32" and not args.no_exact:
        # the same step with exact fp32 MFMA contractions, for reference beside the headline
        conv_layers.set_math("f32")
        for _ in range(2):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt32 = time.perf_counter() - t0
        conv_layers.set_math(conv_math)
        out["exact_fp32"] = {"value": round(args.batch * args.steps / dt32, 3), "unit": "images/sec",
                             "ms_per_step": round(1e3 * dt32 / args.steps, 3), "dtype": "f32",
                             "note": "JTSM_CONV_MATH=f32: v_mfma_f32_32x32x2_f32 contractions, same model, same step"}
    if world > 1:
        torch.distributed.barrier()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_leg(args.size, args.proposals)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
