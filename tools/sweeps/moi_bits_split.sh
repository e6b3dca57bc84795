# which part of the fused bit-table launch is its long pole: the launch timed with parts left out (JTSM_MOI_BITS_SKIP bit l =
# level l's cell bits, bit 4 = the roi bits; JTSM_MOI_SORT=0 = no ordering block)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "0 1" "1 1" "2 1" "4 1" "8 1" "16 1" "31 1" "31 0"; do
set -- $cfg
export JTSM_MOI_BITS_SKIP=$1 JTSM_MOI_SORT=$2
rm -rf gpurun_out/pb; timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/pb -o p --output-format csv -- python tools/sweeps/moi_fwd_ab.py > gpurun_out/pb.log 2>&1
echo "skip $1 sort $2: $(python tools/sweeps/pool_seq.py gpurun_out/pb/p_kernel_trace.csv | grep -E 'moi_bits')"
done
