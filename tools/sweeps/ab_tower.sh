set -e
F="--no-config4 --no-exact --no-cpu-baseline --steps 20"
JTSM_MASK_TOWER=0 timeout -k 10 300 python bench.py $F > gpurun_out/ab_tower0.json 2> gpurun_out/ab.err
JTSM_MASK_TOWER=1 timeout -k 10 300 python bench.py $F > gpurun_out/ab_tower1.json 2>> gpurun_out/ab.err
JTSM_MASK_TOWER=0 timeout -k 10 300 python bench.py $F > gpurun_out/ab_tower0b.json 2>> gpurun_out/ab.err
JTSM_MASK_TOWER=1 timeout -k 10 300 python bench.py $F > gpurun_out/ab_tower1b.json 2>> gpurun_out/ab.err
