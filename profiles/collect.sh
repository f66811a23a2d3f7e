#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's `roofline` object on the GPU
# box (run through gpurun from the repo root):
#   stats : rocprofv3 --kernel-trace --stats of the SAME command the driver runs
#   fetch / write : HBM byte counters, one --pmc pass each (FETCH_SIZE takes 3 TCC
#           slots, WRITE_SIZE 2: MI355X_MICROARCH.md "rocprofv3 PMC slots"), with a
#           calibration launch of known byte count first (bench.py --calibrate)
# Outputs land in gpurun_out/prof_*; profiles/summarize.py turns them into
# profiles/rNN_*.json and profiles/hbm_traffic.json.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_stats -o stats -- \
    python $REPO/bench.py > $OUT/prof_stats_bench.json 2> $OUT/prof_stats.err
PMC_ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-fused --calibrate"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv --kernel-include-regex reg_eval_points \
    -d $OUT/prof_fetch -o fetch -- python $REPO/bench.py $PMC_ARGS \
    > $OUT/prof_fetch_bench.json 2> $OUT/prof_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv --kernel-include-regex reg_eval_points \
    -d $OUT/prof_write -o write -- python $REPO/bench.py $PMC_ARGS \
    > $OUT/prof_write_bench.json 2> $OUT/prof_write.err
cd $REPO
find gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write -name '*.csv' | head -40
python profiles/summarize.py --round ${ROUND:-01} || true
