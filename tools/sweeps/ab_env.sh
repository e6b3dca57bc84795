# usage: bash tools/sweeps/ab_env.sh VAR   -> bench with VAR=0 / VAR=1 twice each, same box
set -e
V=$1
F="--no-config4 --no-exact --no-cpu-baseline --steps 20"
for r in a b; do
  for x in 0 1; do
    env $V=$x timeout -k 10 300 python bench.py $F > gpurun_out/ab_${V}_${x}${r}.json 2>> gpurun_out/ab.err
  done
done
