set -e
for v in 9216 6000 4000 2500; do
  JTSM_X3_MIN_WORK=$v timeout -k 10 300 python tools/sweeps/layer_table.py --min-us 0 > gpurun_out/mw_$v.txt 2>&1
done
