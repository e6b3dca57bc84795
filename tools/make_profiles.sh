#!/bin/bash
# Produce the round's measurement artefacts on the GPU box (run through gpurun from the repo root), in stages that
# each fit one gpurun call:   bash tools/make_profiles.sh pmc | prof | bench | side
#   gpurun_out/art/{pmc/{fetch,write,mfma}, pmc_traffic.json, pmc_mfma.json}                                  (pmc)
#   gpurun_out/art/{kernel_stats.csv (headline workload), kernel_stats_uniform.csv (round-1 workload)}         (prof)
#   gpurun_out/art/{bench.json, bench_detail.json}                                                             (bench)
#   gpurun_out/art/{stock*.json, inference.json, inference_kernel_stats.csv, input_pipeline.json, config1.json} (side)
# Copy what should be judged into profiles/ (named per round) afterwards: gpurun_out/ is scratch.
# Counter passes are separate runs with --kernel-trace only (no other trace domain), as the pool requires.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
A=gpurun_out/art
mkdir -p $A
FAST="--no-cpu-baseline --no-roofline --no-exact --no-config4"
PMCF="--no-cpu-baseline --no-roofline --no-exact"   # (with the configs[4] fp16 leg: its kernels' traffic rows)
case "$1" in
pmc)
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $A/pmc/fetch -o p --output-format csv -- python bench.py --steps 2 --warmup 1 $FAST > $A/pmc_fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $A/pmc/write -o p --output-format csv -- python bench.py --steps 2 --warmup 1 $FAST > $A/pmc_write.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $A/pmc4/fetch -o p --output-format csv -- python bench.py --steps 2 --warmup 1 $PMCF > $A/pmc4_fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $A/pmc4/write -o p --output-format csv -- python bench.py --steps 2 --warmup 1 $PMCF > $A/pmc4_write.log 2>&1 &&
python tools/pmc_traffic.py $A/pmc profiles/pmc_traffic.json $A/pmc4 && cp profiles/pmc_traffic.json $A/pmc_traffic.json &&
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $A/pmc/mfma -o p --output-format csv -- python bench.py --steps 2 --warmup 1 $FAST --launch-sequence $A/launch_sequence.json > $A/pmc_mfma.log 2>&1 &&
(cd tools && python pmc_mfma.py ../$A/pmc/mfma ../profiles/pmc_mfma.json ../$A/launch_sequence.json) && cp profiles/pmc_mfma.json $A/pmc_mfma.json
rm -rf $A/pmc/fetch $A/pmc/write $A/pmc/mfma $A/pmc4   # (the raw per-dispatch tables are tens of MB; the two JSON files are what is kept)
;;
prof)
# (--one-stream: a kernel's duration beside another queue's kernels measures the sharing; the roofline leg measures on
#  one stream too, so its averages agree with this table.  kernel_stats_streams.csv: the default run, durations overlap)
rocprofv3 --kernel-trace --stats -d $A/prof -o r --output-format csv -- python bench.py --one-stream --steps 10 --warmup 3 --no-cpu-baseline --no-exact --no-config4 > $A/prof_bench.json 2> $A/prof.log &&
cp $A/prof/r_kernel_stats.csv $A/kernel_stats.csv &&
rocprofv3 --kernel-trace --stats -d $A/prof_s -o r --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact --no-config4 --no-roofline > $A/prof_s_bench.json 2> $A/prof_s.log &&
cp $A/prof_s/r_kernel_stats.csv $A/kernel_stats_streams.csv &&
rocprofv3 --kernel-trace --stats -d $A/prof_u -o r --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact --no-config4 --no-roofline --cluster 0 > $A/prof_u_bench.json 2> $A/prof_u.log &&
cp $A/prof_u/r_kernel_stats.csv $A/kernel_stats_uniform.csv
rm -rf $A/prof $A/prof_u $A/prof_s
;;
bench)
python bench.py --steps 20 --warmup 5 > $A/bench.json 2> $A/bench.err && cp gpurun_out/bench_detail.json $A/bench_detail.json
;;
side)
python tools/stock_baseline.py --cluster 0 --warmup 25 > $A/stock_uniform.json 2> $A/stock_uniform.err &&
rocprofv3 --kernel-trace --stats -d $A/infprof -o r --output-format csv -- python tools/bench_inference.py --steps 10 --warmup 3 > $A/infprof_bench.json 2> $A/infprof.log &&
cp $A/infprof/r_kernel_stats.csv $A/inference_kernel_stats.csv && rm -rf $A/infprof &&
python tools/bench_inference.py --steps 20 --warmup 5 --cpu-baseline > $A/inference.json 2> $A/inference.err &&
python tools/bench_input_pipeline.py > $A/input_pipeline.json 2> $A/input_pipeline.err &&
python tools/bench_config1.py > $A/config1.json 2> $A/config1.err &&
python tools/stock_baseline.py --warmup 25 > $A/stock.json 2> $A/stock.err
;;
*) echo "usage: make_profiles.sh pmc|prof|bench|side"; exit 2;;
esac
echo "exit $?"; ls $A
