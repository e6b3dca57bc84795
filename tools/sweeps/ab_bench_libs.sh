# usage: bash tools/sweeps/ab_bench_libs.sh A.so B.so ... -> bench.py (20 steps, headline leg only) with the built library, each
# alternative swapped in, and the built one again (GPU box copy only; the tree's library is restored at the end)
F="--no-config4 --no-exact --no-cpu-baseline --no-roofline --steps 20"
cp jtsm_amd/lib/libjtsm_hip.so /tmp/libjtsm_keep.so
timeout -k 10 300 python bench.py $F > gpurun_out/abl_base_a.json 2>> gpurun_out/abl.err
for L in "$@"; do
  cp $L jtsm_amd/lib/libjtsm_hip.so
  timeout -k 10 300 python bench.py $F > gpurun_out/abl_$(basename $L .so).json 2>> gpurun_out/abl.err
done
cp /tmp/libjtsm_keep.so jtsm_amd/lib/libjtsm_hip.so
timeout -k 10 300 python bench.py $F > gpurun_out/abl_base_b.json 2>> gpurun_out/abl.err
for f in gpurun_out/abl_*.json; do python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], d['ms_per_step'])" $f; done
