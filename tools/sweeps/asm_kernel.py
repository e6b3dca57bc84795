"""Instruction census and main-loop listing of one kernel in a hipcc --save-temps .s file.
    python tools/sweeps/asm_kernel.py FILE.s MANGLED_SUBSTRING [--loop]"""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l)
end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i])
body = lines[start:end]
for l in body:
    if any(k in l for k in (".amdhsa_next_free_vgpr", ".amdhsa_accum_offset", "scratch_en", ".amdhsa_group_segment_fixed_size",
                            "private_segment_fixed_size")):
        print(l.strip())
cnt = collections.Counter(l.strip().split()[0] for l in body if l.strip() and not l.strip().startswith((";", ".")))
for k, v in cnt.most_common(40):
    print("%-40s %d" % (k, v))
if "--loop" in sys.argv:
    # the basic block with the most MFMAs
    blocks, cur = [], []
    for l in body:
        if re.match(r"^\.LBB\d+_\d+:", l):
            blocks.append(cur)
            cur = []
        cur.append(l)
    blocks.append(cur)
    best = max(blocks, key=lambda b: sum("v_mfma" in x for x in b))
    for l in best:
        t = l.strip()
        if t and not t.startswith(";"):
            print(t.split(";")[0].rstrip())
