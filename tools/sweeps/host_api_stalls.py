"""Long host-side HIP API calls of a bench.py run (rocprofv3 --hip-trace --kernel-trace ... csv): which calls block the
enqueueing thread, how often per step and for how long.
    python tools/sweeps/host_api_stalls.py DIR [steps] [min_us]"""
import collections, csv, glob, sys

d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
krows = sorted((int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kt)))
sgd = [t for t, n in krows if "sgd_multi_kernel" in n]
t0, t1 = sgd[-steps - 1], sgd[-1]
ht = glob.glob(d + "/**/*hip_api_trace.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
tot = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(ht)):
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if a < t0 or a > t1:
        continue
    us = (b - a) / 1e3
    f = r["Function"]
    tot[f][0] += 1
    tot[f][1] += us
    if us >= min_us:
        e = agg[f]
        e[0] += 1
        e[1] += us
        e[2] = max(e[2], us)
print("host API time inside %d steps, per step:" % steps)
for f, (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:14]:
    print("   %-38s %7.1f calls %9.1f us" % (f, n / steps, us / steps))
print("calls of at least %.0f us, per step:" % min_us)
for f, (n, us, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("   %-38s %7.1f calls %9.1f us   (longest %.0f us)" % (f, n / steps, us / steps, mx))

# which launches are the slow ones: kernel, and whether the launch went to another queue than the one before it
kcols = list(csv.DictReader(open(kt)))
byc = {r["Correlation_Id"]: r for r in kcols}
launches = []
for r in csv.DictReader(open(ht)):
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if r["Function"] in ("hipLaunchKernel", "hipModuleLaunchKernel", "hipExtModuleLaunchKernel") and t0 <= a <= t1:
        k = byc.get(r["Correlation_Id"])
        launches.append((a, (b - a) / 1e3, k["Kernel_Name"] if k else "?", k.get("Queue_Id") if k else None))
launches.sort()
import re
def short(n):
    m = re.search(r"(\w+)(<[^(]*>)?\(", n.replace("(anonymous namespace)::", ""))
    return (m.group(1) + (m.group(2) or ""))[:48] if m else n[:48]
sw = collections.defaultdict(lambda: [0, 0.0])
slow = collections.defaultdict(lambda: [0, 0.0])
prev_q = None
for a, us, name, q in launches:
    key = ("other queue than the launch before" if (prev_q is not None and q != prev_q) else "same queue as the launch before")
    sw[key][0] += 1
    sw[key][1] += us
    if us >= min_us:
        slow[(short(name), key)][0] += 1
        slow[(short(name), key)][1] += us
    prev_q = q
print("launch calls by queue change, per step:")
for k, (n, us) in sw.items():
    print("   %-40s %7.1f calls %9.1f us  (%.1f us per call)" % (k, n / steps, us / steps, us / max(n, 1)))
print("slow launches (>= %.0f us) by kernel, per step:" % min_us)
for (name, key), (n, us) in sorted(slow.items(), key=lambda kv: -kv[1][1])[:25]:
    print("   %-50s %-36s %5.1f calls %8.1f us" % (name, key, n / steps, us / steps))
