# usage: bash tools/sweeps/ab_libs.sh A.so B.so ... -> layer tables with the built library, each alternative swapped in, and the
# built one again (GPU box copy only; the tree's library is restored at the end)
cp jtsm_amd/lib/libjtsm_hip.so /tmp/libjtsm_keep.so
timeout -k 10 300 python tools/sweeps/layer_table.py --min-us 0 > gpurun_out/lt_base_a.txt 2>&1
for L in "$@"; do
  cp $L jtsm_amd/lib/libjtsm_hip.so
  timeout -k 10 300 python tools/sweeps/layer_table.py --min-us 0 > gpurun_out/lt_$(basename $L .so).txt 2>&1
done
cp /tmp/libjtsm_keep.so jtsm_amd/lib/libjtsm_hip.so
timeout -k 10 300 python tools/sweeps/layer_table.py --min-us 0 > gpurun_out/lt_base_b.txt 2>&1
