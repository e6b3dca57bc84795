# Counters of the 256 x 256-tile forward / data-gradient kernel in the shipped build and in the TIMING-ONLY build whose
# activation lo chunks come from the hi chunks' own lines (the request pattern of paired activation planes; wrong values):
#   bash tools/sweeps/build_alt.sh x3_pairedA -DJTSM_TIMING_PAIRED_A     (in the build container)
#   gpurun -- bash tools/sweeps/x3_paired_pmc.sh                          -> gpurun_out/x3_paired_activations.json
O=gpurun_out/x3paired; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
SET1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
SET2="SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD"
for v in shipped pairedA; do
  L=jtsm_amd/lib/libjtsm_hip.so; [ $v = pairedA ] && L=scratch/alt/x3_pairedA.so
  JTSM_HIP_LIB=$L timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/${v}_t -o p --output-format csv -- python tools/sweeps/x3_big.py > $O/${v}_t.log 2>&1 &&
  JTSM_HIP_LIB=$L timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SET1 -d $O/${v}_1 -o p --output-format csv -- python tools/sweeps/x3_big.py > $O/${v}_1.log 2>&1 &&
  JTSM_HIP_LIB=$L timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SET2 -d $O/${v}_2 -o p --output-format csv -- python tools/sweeps/x3_big.py > $O/${v}_2.log 2>&1 || { echo "pass failed: $v"; tail -5 $O/${v}_*.log; exit 1; }
done
python tools/sweeps/x3_paired_pmc.py $O gpurun_out/x3_paired_activations.json
