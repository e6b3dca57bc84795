"""Busy / idle time of the device inside the timed steps of a bench.py run, from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace -d DIR -o t --output-format csv -- python bench.py --steps 10 --warmup 3 ...
    python tools/sweeps/gpu_idle.py DIR [steps]
The union of all kernels' [start, end] intervals (every stream) over the last `steps` steps, delimited by the
optimizer's kernel (sgd_multi_kernel): covered time, gaps, and the gaps' size distribution."""
import csv, glob, sys

d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
sgd = [i for i, r in enumerate(rows) if "sgd_multi_kernel" in r[2]]
assert len(sgd) > steps, len(sgd)
lo, hi = sgd[-steps - 1], sgd[-1]
t0, t1 = rows[lo][1], rows[hi][1]                   # from the end of one optimizer launch to the end of the last
span = [(max(a, t0), min(b, t1), n) for a, b, n in rows if b > t0 and a < t1]
span.sort()
busy, gaps, cur_a, cur_b, last_name = 0, [], span[0][0], span[0][1], span[0][2]
gaps.append(cur_a - t0)
where = []
for a, b, n in span[1:]:
    if a > cur_b:
        busy += cur_b - cur_a
        gaps.append(a - cur_b)
        where.append((a - cur_b, last_name, n))
        cur_a, cur_b, last_name = a, b, n
    else:
        if b > cur_b:
            cur_b, last_name = b, n
busy += cur_b - cur_a
total = t1 - t0
gaps = [g for g in gaps if g > 0]
print("steps %d: %.3f ms per step, device busy %.3f ms (%.1f %%), idle %.3f ms per step in %d gaps per step" % (
    steps, total / steps / 1e6, busy / steps / 1e6, 100.0 * busy / total, (total - busy) / steps / 1e6, len(gaps) // steps))
for lim in (2e3, 5e3, 10e3, 20e3, 50e3, 1e9):
    sel = [g for g in gaps if g <= lim]
    print("  gaps <= %6.0f us: %5d per step, %.3f ms per step" % (lim / 1e3, len(sel) // steps, sum(sel) / steps / 1e6))
big = sorted(gaps, reverse=True)[:5]
print("  largest gaps (us):", [round(g / 1e3, 1) for g in big])
import re
def short(n):
    m = re.search(r"(\w+)(<[^(]*>)?\(", n.replace("(anonymous namespace)::", ""))
    return (m.group(1) + (m.group(2) or ""))[:60] if m else n[:60]
for g, before, after in sorted(where, reverse=True)[:3 * steps]:
    print("    %8.1f us between %s and %s" % (g / 1e3, short(before), short(after)))
