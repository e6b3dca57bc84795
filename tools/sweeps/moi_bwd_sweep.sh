cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for hm in 1024 512 2048; do
export JTSM_MOI_HEAVY_MIN=$hm
rm -rf gpurun_out/pb; timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/pb -o p --output-format csv -- python tools/sweeps/pool_bench2.py moi > gpurun_out/pb.log 2>&1
echo "heavy_min $hm"; python tools/sweeps/pool_seq.py gpurun_out/pb/p_kernel_trace.csv | grep -E "bwd_tiled|bwd_busy|tile_plan|moi_bits|moi_pool_fwd"
done
