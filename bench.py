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


DETAIL_PATH = "gpurun_out/bench_detail.json"
LINE_LIMIT = 4096               # the driver keeps 8 KB of stdout; the final line stays well inside it


def emit(out, detail):
    """Print the ONE JSON line the driver parses (headline keys, `config`, the dominant kernel's `roofline`,
    `cpu_baseline`, flat extra legs) and write everything else — every kernel's roofline, the per-instantiation
    contraction table, the other entry points, the long notes — to gpurun_out/bench_detail.json."""
    line = json.dumps(out)
    if len(line) >= LINE_LIMIT:      # never let detail creep back into the line: drop optional legs, keep the contract
        for k in ("config4_fp16", "round1_workload", "exact_fp32"):
            if k in out and len(line) >= LINE_LIMIT:
                detail.setdefault("moved_from_line", {})[k] = out.pop(k)
                line = json.dumps(out)
    assert len(line) < LINE_LIMIT, "bench line is %d bytes" % len(line)
    if "roofline" not in detail and "config4_fp16" not in detail:   # (a run without its tables: keep the last full file)
        print(line, flush=True)
        return
    try:
        path = os.path.join(ROOT, DETAIL_PATH)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump({"line": out, "detail": detail}, f, indent=1)
    except OSError as e:             # a read-only tree must not cost the measurement
        print("bench detail not written: %s" % e, file=sys.stderr)
    print(line, flush=True)


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
    ap.add_argument("--cluster", type=float, default=1.0,
                    help="fraction of the proposals that are jittered copies of --objects rectangles per image "
                         "(utils/synthetic.py): gives the mask branch a realistic foreground count; 0 = the plain "
                         "uniform recipe of round 1")
    ap.add_argument("--objects", type=int, default=40, help="proposal groups per image (see --cluster)")
    ap.add_argument("--no-config4", action="store_true", help="skip the BASELINE configs[4] extra leg (R101, fp16)")
    ap.add_argument("--one-stream", action="store_true",
                    help="switch the side streams off (semantic head, weight gradients, MOIPool backward's second "
                         "gather) for the WHOLE run: per-kernel durations of a kernel trace then belong to one kernel "
                         "each, as the roofline leg measures them (tools/make_profiles.sh prof)")
    ap.add_argument("--launch-sequence", default=None,
                    help="write the contraction launches of one extra step (kernel, model segment, shape) to this JSON "
                         "file: tools/pmc_mfma.py splits a counter pass of the same command by backbone / FPN / heads")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank code-path rehearsal on a single GPU: every rank uses device 0 and the gloo backend "
                         "(not a measurement)")
    return ap.parse_args()


def build(device, depth=50):
    from model_util import jtsm_cfg
    from jtsm_amd.modeling import build_model

    torch.manual_seed(0)                       # identical random-init weights on every rank
    model = build_model(jtsm_cfg(str(device), depth=depth))
    model.train()
    # random msra weights are not matched to 0..255 inputs; keep activations O(1) (as in the parity tests)
    with torch.no_grad():
        model.backbone.bottom_up.stem.conv1.weight.mul_(1.0 / 64)
        if depth > 50:   # 33 random-init residual blocks double the variance each: damp every block's last norm
            for n, b in model.named_buffers():
                if n.endswith("conv3.norm.weight"):
                    b.mul_(0.3)
    return model


def config4_leg(device, args):
    """BASELINE configs[4] as an EXTRA leg (never the headline: fp16 is narrower than the reference's fp32):
    R101-FPN JTSM panoptic, Cityscapes-shaped 2 x 3 x 1024 x 2048 per GPU, fp16 MFMA path (one fp16 plane per operand,
    v_mfma_f32_32x32x16_f16, fp32 accumulate, fp32 losses), same step (fwd + bwd + SGD)."""
    from jtsm_amd.layers import conv as conv_layers
    from jtsm_amd.utils.synthetic import synthetic_inputs

    old = conv_layers.MATH
    conv_layers.set_math("f16")
    try:
        model = build(device, depth=101)
        opt = make_optimizer(model)
        inputs = synthetic_inputs(1234, batch=args.batch, size=1024, width=2048, proposals=args.proposals,
                                  device=device, cluster=args.cluster, objects=args.objects)

        def step():
            losses = model(inputs)
            total = sum(losses.values())
            total.backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
            return total

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        steps = max(3, args.steps // 2)
        t0 = time.perf_counter()
        for _ in range(steps):
            last = step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        roof = roof_detail = None
        if not args.no_roofline:   # dominant kernel of THIS leg against the fp16 MFMA roof (2.5 PFLOP/s dense)
            streams = set_side_streams(False)         # (as in main(): kernels timed on one stream)
            roof, roof_detail = roofline_leg(step)
            set_side_streams(streams)
            roof["streams"] = "one (side streams off for this leg only)"
            roof.pop("contractions", None)
        out = {"value": round(args.batch * steps / dt, 3), "unit": "images/sec", "ms_per_step": round(1e3 * dt / steps, 3),
               "dtype": "f16", "steps": steps, "final_loss": round(float(last.detach()), 5),
               "foreground_rois_last_step": int(model.roi_heads.aux["fg_classes"].numel()),
               "workload": "BASELINE configs[4] on ONE GPU: R101-FPN JTSM panoptic, %d x 3x1024x2048, %d proposals + %d "
                           "superpixels per image, fp16 MFMA path (fp16 operand planes, fp32 accumulate / losses / "
                           "master weights, gradient planes x 2^%d)" % (args.batch, args.proposals, 32 * 64,
                                                                       conv_layers.GRAD_SHIFT)}
        if roof is not None:
            out["roofline"] = roof
            out["roofline_detail"] = roof_detail
        del model, opt, inputs
        torch.cuda.empty_cache()
        return out
    finally:
        conv_layers.set_math(old)


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
        # (the heads' share of the update as soon as the backward has passed them: solver/build.py: attach_early_heads)
        return SGD(groups, lr=lr, momentum=0.9).attach_early_heads(model)
    return torch.optim.SGD(groups, lr=lr, momentum=0.9)


def _mfma_peak(variant):
    """Dense MFMA roof (TFLOP/s of ALGORITHMIC FLOPs) of a contraction instantiation: split-bf16 issues three bf16
    products per algorithmic product; the fp16 instantiations (NP = 1) one; the igemm_* kernels are fp32 MFMA."""
    if "_x3_" not in variant:
        return FP32_MFMA_PEAK_TFLOPS
    if variant.endswith(",1>") or variant.endswith("<1>"):
        return BF16_MFMA_PEAK_TFLOPS
    return BF16_MFMA_PEAK_TFLOPS / 3.0


def roofline_leg(step_fn):
    """One extra, untimed step with hipEvents around EVERY library launch (contractions: layers/conv.py LAUNCH_LOG,
    with the split-K finishing pass separated by the library's mid-event; everything else: _lib.TIMING).  Returns the
    roofline object of the kernel with the largest time per step — whatever its kind — and `rooflines`: one object
    for every kernel above 2 % of the step's kernel time."""
    from jtsm_amd import _lib
    from jtsm_amd.layers import conv

    conv.LAUNCH_LOG, _lib.TIMING = [], []
    step_fn()
    torch.cuda.synchronize()
    log, conv.LAUNCH_LOG = conv.LAUNCH_LOG, None
    other, _lib.TIMING = _lib.TIMING, None
    per = {}

    def slot(name, bound):
        return per.setdefault(name, {"bound": bound, "launches": 0, "flops": 0.0, "bytes": 0.0, "ms": 0.0, "roof_ms": 0.0,
                                     "hbm_bound_ms": 0.0, "unknown_bytes": 0})

    for variant, flops, span, shape, finish_bytes in log:
        d = slot(str(variant), "mfma")
        d["peak"] = _mfma_peak(str(variant))
        nbytes = float(shape[-1]) if shape is not None else 0.0
        k = span.kernel_ms()            # the contraction kernel alone (library hook), as rocprofv3 reports it
        t_mfma = flops / (d["peak"] * 1e12) * 1e3
        t_hbm = nbytes / (HBM_PEAK_GBS * 1e9) * 1e3
        d["launches"] += 1
        d["flops"] += flops
        d["bytes"] += nbytes
        d["ms"] += k
        d["roof_ms"] += max(t_mfma, t_hbm)          # the tighter roof of THIS launch's shape
        if t_hbm > t_mfma:
            d["hbm_bound_ms"] += k
        fin = max(span.call_ms() - k, 0.0)           # its split-K finishing pass, when there is one
        if finish_bytes > 0:
            f = slot("splitk_finish<4>", "hbm")
            f["launches"] += 1
            f["bytes"] += finish_bytes
            f["ms"] += fin
            f["roof_ms"] += finish_bytes / (HBM_PEAK_GBS * 1e9) * 1e3
            f["hbm_bound_ms"] += fin
    for name, span, nbytes in other:
        d = slot(name, "hbm")
        ms = span.ms()
        d["launches"] += 1
        d["ms"] += ms
        if nbytes is None:
            d["unknown_bytes"] += 1
        else:
            d["bytes"] += nbytes
            d["roof_ms"] += nbytes / (HBM_PEAK_GBS * 1e9) * 1e3
            d["hbm_bound_ms"] += ms
    total_ms = sum(d["ms"] for d in per.values())

    def describe(name, d):
        tflops = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
        gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9 if d["ms"] > 0 else 0.0
        # one instantiation serves shapes on both sides of the ridge: `bound` is the roof that holds for the larger
        # share of the kernel's time
        hbm = d["bound"] == "hbm" or d["hbm_bound_ms"] > 0.5 * d["ms"]
        out = {"kernel": name, "ms_per_step": round(d["ms"], 3), "share_of_kernel_time": round(d["ms"] / total_ms, 4),
               "launches_per_step": d["launches"], "avg_launch_us": round(1e3 * d["ms"] / max(d["launches"], 1), 2)}
        if d["bound"] == "hbm" and d["unknown_bytes"]:
            out.update({"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                        "note": "no algorithmic byte count attached to this entry point"})
            return out
        if hbm:
            out.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(gbs / HBM_PEAK_GBS, 4),
                        "algorithmic_mb_per_launch": round(d["bytes"] / max(d["launches"], 1) / 1e6, 3)})
        else:
            out.update({"bound": "mfma", "achieved": round(tflops, 2), "peak": round(d["peak"], 1), "unit": "TFLOP/s",
                        "frac": round(tflops / d["peak"], 4),
                        "algorithmic_gflop_per_launch": round(d["flops"] / d["launches"] / 1e9, 3),
                        "algorithmic_mb_per_launch": round(d["bytes"] / d["launches"] / 1e6, 2)})
        if d["bound"] == "mfma":
            out.update({"mfma_frac": round(tflops / d["peak"], 4), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
                        "hbm_bound_share_of_time": round(d["hbm_bound_ms"] / d["ms"], 3) if d["ms"] > 0 else None,
                        # per-launch roofline: sum over launches of max(flops/peak_mfma, bytes/peak_hbm) / measured time
                        "roofline_time_frac": round(d["roof_ms"] / d["ms"], 4) if d["ms"] > 0 else None})
        return out

    traffic_tab = {}
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.isfile(pmc):   # HBM bytes per launch from rocprofv3 --pmc passes of this same command
        traffic_tab = json.load(open(pmc)).get("kernels", {})

    def traffic_of(name):
        t = traffic_tab.get(name)
        if not t:
            return None
        return {"hbm_bytes_per_launch": t["hbm_bytes_per_launch"], "source": "profiles/pmc_traffic.json",
                "note": t.get("note", "")}

    ranked = sorted(per.items(), key=lambda kv: -kv[1]["ms"])
    dom_name, D = ranked[0]
    dom = describe(dom_name, D)
    t = traffic_of(dom_name)
    per_launch = D["flops"] / D["launches"] if dom["bound"] == "mfma" else D["bytes"] / max(D["launches"], 1)
    # the contract's object, dominant kernel only (the driver keeps 8 KB of stdout: everything else goes to the
    # detail file).  `traffic`: fabric-side bytes per launch from the committed rocprofv3 PMC passes, or null.
    compact = {"bound": dom["bound"], "achieved": dom["achieved"], "peak": dom["peak"], "unit": dom["unit"],
               "frac": dom["frac"], "traffic": t["hbm_bytes_per_launch"] if t else None,
               "kernel": dom_name, "launches_per_step": D["launches"], "avg_launch_us": dom["avg_launch_us"],
               "algorithmic_per_launch": round(per_launch / 1e9, 4),
               "algorithmic_unit": "GFLOP" if dom["bound"] == "mfma" else "GB",
               "share_of_kernel_time": dom["share_of_kernel_time"], "kernel_time_ms_per_step": round(total_ms, 3),
               "traffic_source": "profiles/pmc_traffic.json" if t else None}
    detail = dict(dom)
    detail["traffic"] = t
    detail["peak_note"] = ("mfma roofs: dense bf16 / fp16 MFMA peak %.0f TFLOP/s (split-bf16 kernels: / 3 bf16 products per "
                           "fp32 product, csrc/conv_x3.h), fp32 MFMA %.1f; algorithmic FLOPs / time.  hbm roof: 8 TB/s; "
                           "algorithmic bytes = every operand read once + the result written once (split-K finishing: every "
                           "slab read once + result, planes and epilogue operands once)" % (BF16_MFMA_PEAK_TFLOPS,
                                                                                            FP32_MFMA_PEAK_TFLOPS))
    rooflines = []
    for name, d in ranked:
        if d["ms"] < 0.02 * total_ms:
            break
        r = describe(name, d)
        r["traffic"] = traffic_of(name)
        rooflines.append(r)
    # the region-pooling entry points, whatever their share (VERDICT r3 item 1 watches them; each is a few launches —
    # bit tables / census / plan + the pooling kernel(s) — timed as one call, traffic: the entry's dominant kernel)
    detail["pooling_entry_points"] = []
    for name in ("jtsm_moi_pool_forward_levels_f32", "jtsm_moi_pool_backward_levels_f32",
                 "jtsm_roi_align_backward_levels_f32", "jtsm_roi_align_forward_level_f32"):
        if name in per and per[name]["ms"] > 0:
            r = describe(name, per[name])
            r["traffic"] = traffic_of(name)
            detail["pooling_entry_points"].append(r)
    contr = {k: v for k, v in per.items() if v["bound"] == "mfma"}
    c_ms = sum(v["ms"] for v in contr.values())
    c_fl = sum(v["flops"] for v in contr.values())
    fin_ms = per.get("splitk_finish<4>", {"ms": 0.0})["ms"]
    detail["kernel_time_ms_per_step"] = round(total_ms, 3)
    detail["rooflines_over_2pct"] = rooflines
    detail["all_contractions"] = {
        "tflops": round(c_fl / ((c_ms + fin_ms) * 1e-3) / 1e12, 2), "ms_per_step": round(c_ms + fin_ms, 3),
        "splitk_finish_ms": round(fin_ms, 3), "gflop_per_step": round(c_fl / 1e9, 1),
        "roofline_time_frac": round(sum(v["roof_ms"] for v in contr.values()) / c_ms, 4) if c_ms > 0 else None,
        "by_kernel": {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                          "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                          "gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1),
                          "roofline_time_frac": round(v["roof_ms"] / v["ms"], 4)}
                      for k, v in sorted(contr.items()) if v["ms"] > 0}}
    detail["other_entry_points"] = {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                                        "gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if not v["unknown_bytes"] and v["ms"] > 0 else None}
                                    for k, v in ranked if v["bound"] == "hbm" and v["ms"] >= 0.01}
    compact["contractions"] = {"tflops": detail["all_contractions"]["tflops"],
                               "ms_per_step": detail["all_contractions"]["ms_per_step"],
                               "splitk_finish_ms": detail["all_contractions"]["splitk_finish_ms"],
                               "gflop_per_step": detail["all_contractions"]["gflop_per_step"]}
    return compact, detail


def _cpu_oracle_rate(size, proposals, threads, warmup, iters):
    """images/sec of the torch-CPU oracle's training step (forward + backward, no optimizer) on ONE image."""
    from oracle import model as OM

    old = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        p = OM.init_params(0, input_gain=1.0 / 64)
        for n in OM.trainable_names(p):
            p[n].requires_grad_(True)
        b = OM.synthetic_batch(1234, B=1, size=size, R=proposals, sp_block=32)
        times = []
        for it in range(warmup + iters):
            for n in OM.trainable_names(p):
                p[n].grad = None
            t0 = time.perf_counter()
            losses = OM.forward_losses(p, b)
            sum(losses.values()).backward()
            if it >= warmup:
                times.append(time.perf_counter() - t0)
        return len(times) / sum(times), times
    finally:
        torch.set_num_threads(old)


def cpu_baseline_leg(size, proposals):
    """The CPU oracle (oracle/model.py, a torch-CPU port of the reference's arithmetic) on a bounded sample of the
    same workload, as SURVEY §8d asks: warmed up, several timed iterations, at all host threads AND at one.
    `value` is the all-threads rate on one FULL-SIZE image (1 warm-up + 3 timed steps); the single-thread rate is
    measured on a quarter-size sample (half the image side, a quarter of the proposals: a full-size step takes ~1
    minute on one thread) and reported both as measured and scaled by the 4x work ratio."""
    cores = torch.get_num_threads()
    rate, times = _cpu_oracle_rate(size, proposals, cores, 1, 3)
    small = (size // 2, proposals // 4)
    rate1, times1 = _cpu_oracle_rate(small[0], small[1], 1, 1, 3)
    return {"value": round(rate, 4), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "torch-CPU oracle training step (fwd+bwd, no optimizer), 1 warm-up + 3 timed: all %d threads on 1 "
                      "image %dx%d / %d proposals (%s s); pooling ops single-threaded C like the reference"
                      % (cores, size, size, proposals, "/".join("%.1f" % t for t in times)),
            "single_thread": {"value": round(rate1 / 4.0, 4), "cores": 1,
                              "measured": round(rate1, 4),
                              "sample": "same step, 1 thread, 1 image %dx%d / %d proposals (%s s); value = measured / 4 "
                                        "(work ratio to the full-size image)"
                                        % (small[0], small[0], small[1], "/".join("%.1f" % t for t in times1))}}


def set_side_streams(on):
    """The Python-level side streams (layers/conv.py: weight gradients; meta_arch/mcnn.py: semantic head) on or off;
    returns the previous setting."""
    from jtsm_amd.layers import conv
    from jtsm_amd.modeling.meta_arch import mcnn
    prev = (conv.WGRAD_STREAM, mcnn.SEM_SIDE_STREAM)
    conv.WGRAD_STREAM, mcnn.SEM_SIDE_STREAM = (on, on) if isinstance(on, bool) else on
    return prev


def main():
    args = parse()
    if args.one_stream:
        os.environ["JTSM_MOI_BWD_STREAMS"] = "0"     # (read by the library at its first pooling backward)
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
    if args.one_stream:
        set_side_streams(False)
    model = build(device)
    inputs = synthetic_inputs(1234 + rank, batch=args.batch, size=args.size, proposals=args.proposals, device=device,
                              cluster=args.cluster, objects=args.objects)
    # one process per GPU; this repo's own bucketed reduce-scatter + all-gather over RCCL, overlapped with the backward
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

    out, detail = None, {}
    if rank == 0:
        ims = args.batch * world * args.steps / dt
        out = {
            "metric": "images/sec training, R50-FPN JTSM panoptic, 2x1024x1024, 1/2/4/8 GPU",
            "value": round(ims, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": conv_math, "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[2]: JTSM panoptic R50-FPN, %d x 3x%dx%d per GPU, %d proposals + %d "
                            "superpixels per image; MIL + 4 OICR + 2 mask heads + sem-seg; fwd + bwd + gradient "
                            "exchange + SGD; %d%% of proposals jittered copies of %d rectangles per image" % (
                                args.batch, args.size, args.size, args.proposals, (args.size // 32) ** 2,
                                round(100 * args.cluster), args.objects),
                "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                "math": ("split-bf16 (3 bf16 MFMA products, fp32 accumulate; <= 6e-6 rel per layer; bar 1e-4); "
                         "see exact_fp32") if conv_math == "bf16x3" else conv_math,
                "final_loss": round(loss_value, 5), "lr": 1e-7,
                "foreground_rois_last_step": int(model.roi_heads.aux["fg_classes"].numel()),
                "detail": DETAIL_PATH,
            },
        }
        detail["config"] = {
            "substitutions": "grabCut -> the reference's own superpixel-evidence masks (object_evidence, "
                             "roi_heads_jtsm.py:1928-1994); polygon encoding of masks skipped (bitmasks); dropout on",
            "weights": "random init (msra/xavier as the reference), FrozenBN identity, stem x1/64",
            "math": "contractions in split-bf16: fp32 operands -> bf16 hi+lo planes, a_lo*b_hi + a_hi*b_lo + a_hi*b_hi "
                    "on v_mfma_f32_32x32x16_bf16 with fp32 accumulate; measured error <= 6e-6 relative per layer "
                    "against fp64 (bar 1e-4); everything else fp32. JTSM_CONV_MATH=f32 selects exact fp32 MFMA",
            "proposals": "%d %% of the proposals are jittered copies (each edge +-12 %%) of %d rectangles per image, the way "
                         "real proposal sets crowd around regions, the rest uniform" % (round(100 * args.cluster), args.objects),
            "torch_device_ops": ["autograd gradient-accumulation adds", "sort glue of the label path",
                                 "RCCL collectives (N > 1)"]}
    if args.launch_sequence and world == 1:
        # the contraction launches of ONE step, in order, with the part of the model they belong to: what
        # tools/pmc_mfma.py aligns a counter pass of this command with (every step launches the same sequence)
        conv_layers.LAUNCH_LOG = []
        step()
        torch.cuda.synchronize()
        seq, conv_layers.LAUNCH_LOG = conv_layers.LAUNCH_LOG, None
        with open(args.launch_sequence, "w") as f:
            json.dump([{"kernel": str(v), "segment": getattr(v, "segment", None), "shape": list(shape[:-1])}
                       for v, _, _, shape, _ in seq], f)
    if not args.no_roofline:
        # every rank runs the extra (untimed) step — it contains the gradient all-reduce — rank 0 reports it
        # with the side streams OFF for this one step: beside another queue's kernels a launch's duration measures
        # how the two share the chip, not the kernel (the 256 x 256-tile data gradient: 234 us alone, 312 us beside the
        # semantic head's kernels); `value` above is the step WITH them
        streams = set_side_streams(False)
        roof, roof_detail = roofline_leg(step)
        set_side_streams(streams)
        roof["streams"] = "one (side streams off for this leg only; see --one-stream)"
        if rank == 0:
            out["roofline"] = roof
            detail["roofline"] = roof_detail
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
                             "ms_per_step": round(1e3 * dt32 / args.steps, 3), "dtype": "f32"}
    if world == 1 and args.cluster > 0 and not args.no_exact:
        # continuity with round 1: the same model and step on the uniform proposal recipe (foreground rois ~ 7)
        inputs_u = synthetic_inputs(1234 + rank, batch=args.batch, size=args.size, proposals=args.proposals, device=device)
        inputs, keep = inputs_u, inputs
        for _ in range(2):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dtu = time.perf_counter() - t0
        out["round1_workload"] = {"value": round(args.batch * args.steps / dtu, 3), "unit": "images/sec",
                                  "ms_per_step": round(1e3 * dtu / args.steps, 3), "dtype": conv_math,
                                  "foreground_rois_last_step": int(model.roi_heads.aux["fg_classes"].numel())}
        inputs = keep
    if world == 1 and not args.no_config4:
        del net, opt, model, inputs
        torch.cuda.empty_cache()
        c4 = config4_leg(device, args)
        detail["config4_fp16"] = dict(c4)
        out["config4_fp16"] = {k: c4[k] for k in ("value", "unit", "ms_per_step", "dtype", "steps",
                                                  "foreground_rois_last_step") if k in c4}
        if "roofline" in c4:
            out["config4_fp16"]["roofline"] = c4["roofline"]
    if world > 1:
        torch.distributed.barrier()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_leg(args.size, args.proposals)
    if rank == 0:
        emit(out, detail)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
