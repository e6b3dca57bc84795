# usage: bash tools/sweeps/ab_layers.sh VAR -> layer tables with VAR=0 and VAR=1 on the same box
V=$1
env $V=0 timeout -k 10 300 python tools/sweeps/layer_table.py --min-us 0 > gpurun_out/lt_${V}_0.txt 2>&1
env $V=1 timeout -k 10 300 python tools/sweeps/layer_table.py --min-us 0 > gpurun_out/lt_${V}_1.txt 2>&1
