"""MFMA pipe utilisation of every contraction kernel from one rocprofv3 counter pass of the bench command:

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d <dir> -o p \\
        --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-exact
    python tools/pmc_mfma.py <dir> profiles/pmc_mfma.json

SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs (256 CUs x 4), GRBM_GUI_ACTIVE over the 8 XCDs, both in
shader-clock cycles, so   util = MFMA_BUSY / (1024 * GUI_ACTIVE / 8)   is the busy fraction of the matrix pipes at
the clock the kernel actually ran at, and   clock = GUI_ACTIVE / 8 / duration.   Calibration: the fc6 forward GEMM
runs 415 TFLOP/s fp32-equivalent = 3 x 415 bf16 TFLOP/s = 50 % of the 2.5 PFLOP/s peak quoted at 2.4 GHz; the counters
give 61 % at a measured 1.96 GHz, i.e. 0.61 x 1.96 / 2.4 = 50 % — the two agree."""
import collections
import csv
import glob
import json
import sys

from pmc_traffic import label

SIMDS, XCDS = 1024, 8


def by_segment(disp, dur, seq_path):
    """Split the contraction dispatches by model segment.  `seq_path` (bench.py --launch-sequence): the contraction
    launches of ONE step in order, each with its segment.  Every step of the profiled command launches the same
    sequence, so for each kernel name the i-th dispatch of the pass is the (i mod n)-th launch of that name in the list."""
    seq = json.load(open(seq_path))
    per_name = collections.defaultdict(list)
    for e in seq:
        per_name[e["kernel"]].append(e["segment"] or "heads")
    seen = collections.defaultdict(int)
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    unmatched = 0
    # aligned from the END of the pass: the listed step is the process's last one, so the last n dispatches of a name ARE
    # its n launches of the list; earlier steps repeat them (a first step that launches something once only — plan
    # building, a different foreground count — cannot shift the later ones)
    for i in sorted(disp, key=lambda d: -int(d)):
        v = disp[i]
        k = label(v["name"])
        if not (k and k.startswith("igemm")):
            continue
        segs = per_name.get(k)
        if not segs:
            unmatched += 1
            continue
        seg = segs[len(segs) - 1 - seen[k] % len(segs)]
        seen[k] += 1
        a = agg[seg]
        a[0] += 1
        a[1] += v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        a[2] += v.get("GRBM_GUI_ACTIVE", 0.0)
        a[3] += dur.get(i, 0)
    tn = sum(a[3] for a in agg.values()) or 1
    out = {seg: {"launches_profiled": n, "mfma_util": round(busy / (SIMDS * gui / XCDS), 4) if gui else None,
                 "clock_ghz": round(gui / XCDS / ns, 3) if ns else None, "share_of_contraction_time": round(ns / tn, 4)}
           for seg, (n, busy, gui, ns) in sorted(agg.items())}
    out["_unmatched_dispatches"] = unmatched
    # a sanity check of the alignment: every kernel name's dispatch count must be a multiple of its per-step count
    out["_aligned"] = all(seen[k] % len(per_name[k]) == 0 for k in seen)
    if not out["_aligned"]:
        # (dispatches of the pass, launches of the listed step: a warm-up step with another foreground count runs the mask
        #  heads on other instantiations; the split above is aligned from the END of the pass, where the steps are alike)
        out["_misaligned"] = {k: [seen[k], len(per_name[k])] for k in seen if seen[k] % len(per_name[k])}
    return out


def main():
    root, out = sys.argv[1], sys.argv[2]
    seq_path = sys.argv[3] if len(sys.argv) > 3 else None
    cc = glob.glob(root + "/**/*counter_collection.csv", recursive=True)[0]
    kt = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))}
    disp = collections.defaultdict(dict)
    for r in csv.DictReader(open(cc)):
        d = disp[r["Dispatch_Id"]]
        d[r["Counter_Name"]] = float(r["Counter_Value"])
        d["name"] = r["Kernel_Name"]
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for i, v in disp.items():
        k = label(v["name"])
        if k and k.startswith("igemm"):
            a = agg[k]
            a[0] += 1
            a[1] += v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
            a[2] += v.get("GRBM_GUI_ACTIVE", 0.0)
            a[3] += dur.get(i, 0)
    kernels = {}
    for k, (n, busy, gui, ns) in sorted(agg.items()):
        kernels[k] = {"launches_profiled": n, "avg_us": round(ns / n / 1e3, 1),
                      "mfma_util": round(busy / (SIMDS * gui / XCDS), 4) if gui else None,
                      "clock_ghz": round(gui / XCDS / ns, 3) if ns else None,
                      "share_of_contraction_time": 0.0}
    tb, tg, tn = (sum(a[i] for a in agg.values()) for i in (1, 2, 3))
    for k, a in agg.items():
        kernels[k]["share_of_contraction_time"] = round(a[3] / tn, 4)
    segments = by_segment(disp, dur, seq_path) if seq_path else None
    json.dump({"by_segment": segments, "by_segment_note": "backbone = ResNet stages (the north star's 'backbone convs'), fpn = lateral / "
               "output convolutions, heads = box head, predictors, mask towers, semantic head; split through bench.py "
               "--launch-sequence (tools/pmc_mfma.py: by_segment)", "command": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- "
                          "python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-exact",
               "formula": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs); "
                          "clock = GRBM_GUI_ACTIVE / 8 / duration",
               "note": "clock_ghz is meaningful for launches of >= 150 us only: the counter window of a short dispatch "
                       "is wider than its timestamps",
               "all_contractions": {"mfma_util": round(tb / (SIMDS * tg / XCDS), 4),
                                    "clock_ghz": round(tg / XCDS / tn, 3)},
               "kernels": kernels}, open(out, "w"), indent=1)
    print("wrote", out, "all contractions: util %.3f at %.2f GHz" % (tb / (SIMDS * tg / XCDS), tg / XCDS / tn))


if __name__ == "__main__":
    main()
