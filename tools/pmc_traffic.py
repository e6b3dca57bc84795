"""Turn two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE — they do not fit one pass, MI355X_MICROARCH.md
§HBM) of the bench command into profiles/pmc_traffic.json: HBM-side bytes per launch for every contraction
kernel, keyed by the label bench.py's roofline leg uses.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d <dir>/fetch -o p --output-format csv -- python bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d <dir>/write -o p --output-format csv -- python bench.py ...
    python tools/pmc_traffic.py <dir> profiles/pmc_traffic.json

gfx950 correction (same guide): FETCH_SIZE tallies a wide coalesced read's 128-byte requests at 64 bytes, so it
is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores (the wide epilogue) and for float atomics.
"""
import collections
import csv
import glob
import json
import re
import sys

ROLE = {"0": "FWD", "1": "DGRAD", "2": "WGRAD"}


# non-contraction kernels that bench.py reports under their C-ABI entry point (the entry's dominant kernel)
ENTRY = (("splitk_finish<4", "splitk_finish<4>"),   # <4, 1> and <4, 16> (threads per piece) book as one
         ("moi_pool_fwd_levels", "jtsm_moi_pool_forward_levels_f32"), ("moi_pool_fwd_rows", "jtsm_moi_pool_forward_levels_f32"),
         ("moi_pool_bwd_tiled", "jtsm_moi_pool_backward_levels_f32"), ("align_bwd_gather", "jtsm_roi_align_backward_levels_f32"),
         ("relu_bwd_split_kernel", "jtsm_relu_backward_split_f32"), ("channel_sum4_kernel", "jtsm_channel_sum_ws_f32"),
         ("split_bf16_kernel", "jtsm_split_bf16_f32"), ("sgd_multi_kernel", "jtsm_sgd_momentum_multi_f32"),
         ("ce_up_bwd_kernel", "jtsm_semseg_ce_backward_f32"),
         ("channel_sum_planes_kernel", "jtsm_channel_sum_planes_multi"),
         ("paste_crop_targets_kernel", "jtsm_paste_crop_targets_f32"))


def label(name):
    """rocprof kernel name -> the label bench.py's roofline leg uses.  The bf16x3 / fp16 kernels carry the plane count
    as their LAST template argument (2 = split-bf16: dropped from the label; 1 = fp16: kept as ",1")."""
    def np_suffix(v):
        return "" if v in (None, "2") else "," + v

    m = re.search(r"igemm_x3_wgrad_halo_group_kernel(?:<(\d+)>)?", name)   # grouped launches (conv_x3.h: X3Group)
    if m:
        return "igemm_x3_wgrad_halo_group_kernel" + ("<1>" if m.group(1) == "1" else "")
    m = re.search(r"igemm_x3_wgrad_group_kernel<(\d+), (\d+), (\d+), (\d+), (\d+)(?:, (\d+))?>", name)
    if m:
        g = m.groups()
        return "igemm_x3_wgrad_group_kernel<%s,%s,%s,%s,%s%s>" % (g[:5] + (np_suffix(g[5]),))
    m = re.search(r"igemm_x3_wgrad_halo_kernel(?:<(\d+)(?:, (?:true|false))?>)?", name)   # <NP, BIAS>
    if m:
        return "igemm_x3_wgrad_halo_kernel" + ("<1>" if m.group(1) == "1" else "")
    m = re.search(r"igemm_x3_wgrad_kernel<(\d+), (\d+), (\d+), (\d+), (\d+)(?:, (\d+))?(?:, (?:true|false))?>", name)
    if m:
        g = m.groups()
        return "igemm_x3_wgrad_kernel<%s,%s,%s,%s,%s%s>" % (g[:5] + (np_suffix(g[5]),))
    m = re.search(r"igemm_x3_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)(?:, (\d+))?>", name)
    if m:
        g = m.groups()
        return "igemm_x3_kernel<%s,%s,%s,%s,%s,%s%s>" % ((ROLE[g[0]],) + g[1:6] + (np_suffix(g[6]),))
    m = re.search(r"igemm_x3_halo_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)(?:, (\d+))?>", name)
    if m:
        g = m.groups()
        return "igemm_x3_halo_kernel<%s,%s,%s,%s,%s,%s%s>" % ((ROLE[g[0]],) + g[1:6] + (np_suffix(g[6]),))
    m = re.search(r"igemm_dma_kernel<(\d+), (\d+), (\d+), (\d+)>", name)
    if m:
        g = m.groups()
        return "igemm_dma_kernel<%s,%s,%s,%s>" % ((ROLE[g[0]],) + g[1:])
    m = re.search(r"igemm_kernel<(\d+), (\d+), (\d+)>", name)
    if m:
        g = m.groups()
        return "igemm_kernel<%s,%s,%s>" % ((ROLE[g[0]],) + g[1:])
    for needle, entry in ENTRY:
        if needle in name:
            return entry
    return None


def per_kernel(directory, counter):
    files = glob.glob(directory + "/**/*counter_collection.csv", recursive=True)
    if not files:
        raise SystemExit("no counter_collection.csv under " + directory)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != counter:
            continue
        k = label(r["Kernel_Name"])
        if k:
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    return agg


def is_fp16(k):
    """Labels of the fp16 instantiations (the configs[4] leg): NP = 1 as the last template argument."""
    return k.endswith(",1>") or k.endswith("<1>")


def main():
    """pmc_traffic.py ROOT OUT [ROOT4]: ROOT = passes of the headline step alone; ROOT4 (optional) = passes of a run
    that includes the configs[4] leg, from which ONLY the fp16 kernels' rows are taken — entry points without a
    template argument to tell the legs apart (the pooling calls: other maps, other label widths) would otherwise be
    averaged over both."""
    root, out = sys.argv[1], sys.argv[2]
    fetch, write = per_kernel(root + "/fetch", "FETCH_SIZE"), per_kernel(root + "/write", "WRITE_SIZE")
    if len(sys.argv) > 3:
        f4, w4 = per_kernel(sys.argv[3] + "/fetch", "FETCH_SIZE"), per_kernel(sys.argv[3] + "/write", "WRITE_SIZE")
        fetch.update({k: v for k, v in f4.items() if is_fp16(k)})
        write.update({k: v for k, v in w4.items() if is_fp16(k)})
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        nf, f = fetch.get(k, [0, 0.0])
        nw, w = write.get(k, [0, 0.0])
        fkb = f / nf if nf else 0.0
        wkb = w / nw if nw else 0.0
        kernels[k] = {
            "launches_profiled": max(nf, nw),
            "fetch_size_kb_per_launch": round(fkb, 1),
            "write_size_kb_per_launch": round(wkb, 1),
            "hbm_bytes_per_launch": int(2 * fkb * 1024 + wkb * 1024),
            "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (KB); gfx950 correction: FETCH_SIZE x2 for "
                    "16-B/lane coalesced reads (MI355X_MICROARCH.md HBM section); Infinity-Cache hits are counted, so this "
                    "is fabric-side traffic, an upper bound on HBM bytes",
        }
    json.dump({"command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> -- python bench.py --steps 2 --warmup 1 "
                          "--no-cpu-baseline --no-roofline --no-exact --no-config4 (the fp16 kernels' rows — \",1\" labels — from a second pair of passes that includes the configs[4] leg)", "kernels": kernels}, open(out, "w"), indent=1)
    print("wrote", out, len(kernels), "kernels")


if __name__ == "__main__":
    main()
