"""Skeleton (waits, loads, LDS traffic, MFMA runs, branches) of one kernel's K loop in a hipcc --save-temps .s file.
    python tools/sweeps/asm_loop.py FILE.s MANGLED_SUBSTRING"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l)
end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i])
body = lines[start:end]
for l in body:
    if "next_free_vgpr" in l or "accum_offset" in l or "private_segment_fixed" in l:
        print(l.strip())
keys = ("s_waitcnt", "global_load", "ds_write", "s_barrier", "v_mfma", "ds_read", "s_cbranch", "s_branch", "scratch_",
        "v_accvgpr", "buffer_load")
out = []
for i, l in enumerate(body):
    t = l.strip()
    if re.match(r"^\.LBB\d+_\d+:", l) or any(k in t for k in keys):
        out.append("%5d %s" % (i, t.split(";")[0][:70]))
first = next(i for i, l in enumerate(out) if "s_barrier" in l)
last = max(i for i, l in enumerate(out) if "v_mfma" in l)
prev, n = None, 0
for l in out[max(0, first - 3):last + 5] + ["0 end"]:
    op = l.split()[1]
    if op.startswith(("v_mfma", "ds_read_b128", "ds_read_b64_tr_b16")):
        if prev == op:
            n += 1
            continue
        if prev:
            print("      %s x%d" % (prev, n))
        prev, n = op, 1
    else:
        if prev:
            print("      %s x%d" % (prev, n))
        prev, n = None, 0
        print(l)
