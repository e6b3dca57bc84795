# usage: bash tools/sweeps/ab_layers2.sh VAR A B -> layer tables with VAR=A and VAR=B on the same box
V=$1
env $V=$2 timeout -k 10 300 python tools/sweeps/layer_table.py --min-us 0 > gpurun_out/lt_${V}_a.txt 2>&1
env $V=$3 timeout -k 10 300 python tools/sweeps/layer_table.py --min-us 0 > gpurun_out/lt_${V}_b.txt 2>&1
