"""Fold the passes of tools/sweeps/x3_paired_pmc.sh into one JSON: per build and per 256 x 256-tile kernel (forward /
data gradient) the launch time, the matrix pipes' busy fraction and the wavefronts' wait / issue fractions."""
import collections, csv, glob, json, re, sys

root, out = sys.argv[1], sys.argv[2]
SIMDS, XCDS = 1024, 8


def table(d, pattern):
    f = glob.glob(d + "/**/" + pattern, recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []


def kernel(name):
    m = re.search(r"igemm_x3_kernel<(\d), 4, 2, 2, 4, 2, 2>", name)
    return {"0": "igemm_x3_kernel<FWD,4,2,2,4,2>", "1": "igemm_x3_kernel<DGRAD,4,2,2,4,2>"}.get(m.group(1)) if m else None


res = {}
for build in ("shipped", "pairedA"):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in table("%s/%s_t" % (root, build), "*kernel_stats.csv"):
        k = kernel(r["Name"])
        if k:
            per[k]["avg_us_kernel_trace"] = round(float(r["AverageNs"]) / 1e3, 1)
            per[k]["launches_traced"] = int(r["Calls"])
    for p in ("1", "2"):
        for r in table("%s/%s_%s" % (root, build, p), "*counter_collection.csv"):
            k = kernel(r["Kernel_Name"])
            if k:
                per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    res[build] = {}
    for k, c in per.items():
        wave = c.get("SQ_WAVE_CYCLES") or 1.0
        gui = c.get("GRBM_GUI_ACTIVE") or 1.0
        res[build][k] = {
            "avg_us_kernel_trace": c.get("avg_us_kernel_trace"), "launches_traced": c.get("launches_traced"),
            "mfma_busy": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (SIMDS * gui / XCDS), 4),
            "wait_inst_any_per_wave_cycle": round(c.get("SQ_WAIT_INST_ANY", 0.0) / wave, 4),
            "wait_inst_lds_per_wave_cycle": round(c.get("SQ_WAIT_INST_LDS", 0.0) / wave, 4),
            "active_inst_vmem_per_wave_cycle": round(c.get("SQ_ACTIVE_INST_VMEM", 0.0) / wave, 4),
            "active_inst_lds_per_wave_cycle": round(c.get("SQ_ACTIVE_INST_LDS", 0.0) / wave, 4),
            "sq_busy_per_gui": round(c.get("SQ_BUSY_CYCLES", 0.0) / gui, 4),
            "raw": {a: b for a, b in c.items() if a.startswith(("SQ_", "GRBM_"))},
        }
json.dump({
    "what": "the 256 x 256-tile bf16x3 forward / data-gradient kernel on tools/sweeps/x3_big.py's three layers (mask-head 3x3 at "
            "255 x 14 x 14 x 256, fc1 4000 x 12544 -> 2048, fc2-sized 4000 x 2048 -> 4096; 10 launches each): the shipped build "
            "against a TIMING-ONLY build (-DJTSM_TIMING_PAIRED_A, wrong values) whose activation lo chunks are fetched from the "
            "hi chunks' own 128-byte lines, i.e. the L2 request pattern of paired activation planes",
    "formulas": "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs); the other fractions are "
                "per SQ_WAVE_CYCLES (summed over the kernel's launches)",
    "commands": "tools/sweeps/x3_paired_pmc.sh (three rocprofv3 passes per build: --kernel-trace --stats; two --pmc sets)",
    "builds": res}, open(out, "w"), indent=1)
for b, ks in res.items():
    for k, v in ks.items():
        print(b, k, {a: c for a, c in v.items() if a != "raw"})
