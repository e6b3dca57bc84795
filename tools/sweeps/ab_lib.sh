# usage: bash tools/sweeps/ab_lib.sh ALT.so -> layer tables with the built library and with ALT.so swapped in (GPU box copy only)
timeout -k 10 300 python tools/sweeps/layer_table.py --min-us 0 > gpurun_out/lt_lib_a.txt 2>&1
cp jtsm_amd/lib/libjtsm_hip.so /tmp/libjtsm_keep.so && cp $1 jtsm_amd/lib/libjtsm_hip.so
timeout -k 10 300 python tools/sweeps/layer_table.py --min-us 0 > gpurun_out/lt_lib_b.txt 2>&1
cp /tmp/libjtsm_keep.so jtsm_amd/lib/libjtsm_hip.so
timeout -k 10 300 python tools/sweeps/layer_table.py --min-us 0 > gpurun_out/lt_lib_a2.txt 2>&1
