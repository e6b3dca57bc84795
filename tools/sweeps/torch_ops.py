"""What PyTorch's own device kernels still cost in one step: reads a rocprofv3 --kernel-trace CSV of
tools/sweeps/step_times.py, takes the last full step (between two SGD launches) and lists the non-library kernels.
    python tools/sweeps/torch_ops.py DIR/p_kernel_trace.csv [--all]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "sgd_multi" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
step = rows[a + 1:b + 1]
t0 = int(step[0]["Start_Timestamp"])
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3


def short(n):
    return re.sub(r"void |at::native::|\(anonymous namespace\)::|jtsm::", "", n)[:100]


busy = sum(dur(r) for r in step)
span = (int(step[-1]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3
tot, cnt = collections.Counter(), collections.Counter()
for i, r in enumerate(step):
    n = r["Kernel_Name"]
    lib = "jtsm::" in n or n.startswith("_ZN4jtsm")
    if not lib:
        tot[short(n)[:70]] += dur(r)
        cnt[short(n)[:70]] += 1
    if "--all" in sys.argv and (not lib or dur(r) > 50):
        print("%.3f %7.1f us g%-8s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, dur(r), r["Grid_Size_X"], short(n)))
print("step: %d launches, %.3f ms of kernels over %.3f ms; non-library: %d launches, %.3f ms" % (
    len(step), busy / 1e3, span / 1e3, sum(cnt.values()), sum(tot.values()) / 1e3))
for k, v in tot.most_common(25):
    print("%8.1f us %3d  %s" % (v, cnt[k], k))
