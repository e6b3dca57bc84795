# MOIPool forward: kernel time + fabric-side fetch bytes with / without the rois' spatial order (run through gpurun)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/moifwd
rm -rf $O; mkdir -p $O
for cfg in "1 0" "1 1" "0 0"; do
  set -- $cfg
  export JTSM_MOI_FWD_ROWS=$1 JTSM_MOI_SORT=$2
  python tools/sweeps/moi_fwd_ab.py 2>&1 | grep "us per forward" | sed "s/^/rows $1 sort $2: /"
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/m$1s$2 -o p --output-format csv -- python tools/sweeps/moi_fwd_ab.py > $O/m$1s$2.log 2>&1 || exit 1
  python - <<PY
import csv, glob, collections
f = glob.glob("$O/m$1s$2/**/p_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "moi_pool_fwd" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        acc[r["Kernel_Name"].split("(")[0][-30:]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    h = len(v) // 2
    print("rows $1 sort $2", k, "FETCH_SIZE x2 MB: clustered %.0f uniform %.0f" % (2 * sum(v[:h]) / h * 1024 / 1e6, 2 * sum(v[h:]) / (len(v) - h) * 1024 / 1e6))
PY
done
