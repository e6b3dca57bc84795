# usage: bash tools/sweeps/ab_env_layers.sh VAR v1 v2 ... -> layer tables with VAR unset, VAR=v1, VAR=v2, ..., unset again (same box)
V=$1; shift
timeout -k 10 300 python tools/sweeps/layer_table.py --min-us 0 > gpurun_out/lt_${V}_base_a.txt 2>&1
for x in "$@"; do
  env $V=$x timeout -k 10 300 python tools/sweeps/layer_table.py --min-us 0 > gpurun_out/lt_${V}_$x.txt 2>&1
done
timeout -k 10 300 python tools/sweeps/layer_table.py --min-us 0 > gpurun_out/lt_${V}_base_b.txt 2>&1
