# MOIPool forward: SQ / TA / TCP / TCC counters per kernel form (run through gpurun from the repo root).
#   bash tools/sweeps/moi_fwd_counters.sh "0 1"      (JTSM_MOI_FWD_ROWS modes)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/moicnt
mkdir -p $O
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
P2="TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"
P3="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
P4="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
for m in ${1:-0 1}; do
  export JTSM_MOI_FWD_ROWS=$m SORT=${SORT:-0}
  i=0
  for P in "$P1" "$P2" "$P3" "$P4"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $P -d $O/m${m}p$i -o p --output-format csv -- python tools/sweeps/moi_fwd_ab.py > $O/m${m}p$i.log 2>&1 || { tail -5 $O/m${m}p$i.log; exit 1; }
  done
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$O/m${m}p*/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "moi_pool_fwd" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("mode $m (clustered half of the calls):")
for k in sorted(acc):
    v = acc[k]; h = len(v) // 2
    print("  %-40s %14.0f" % (k, sum(v[3:h]) / max(h - 3, 1)))
PY
done
