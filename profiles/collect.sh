#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's `roofline` object on the GPU
# box (run through gpurun from the repo root):
#   stats : rocprofv3 --kernel-trace --stats of the SAME command the driver runs
#   rd / fetch / write : HBM byte counters, one --pmc pass each (FETCH_SIZE takes 3 TCC
#           slots, WRITE_SIZE 2: MI355X_MICROARCH.md "rocprofv3 PMC slots"), with
#           calibration launches of known output byte count first (bench.py --calibrate)
# Outputs land in gpurun_out/prof_*; profiles/summarize.py turns them into
# profiles/rNN_*.json and profiles/hbm_traffic.json.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_stats -o stats -- \
    python $REPO/bench.py > $OUT/prof_stats_bench.json 2> $OUT/prof_stats.err
# Counter passes (one --pmc set per run; never combined with hip/hsa/sys traces).  The command
# launches, in order: two calibration dispatches of reg_eval_points_kernel (poses 10 km apart:
# no evaluation finds a reading block, so each writes exactly 36 B per residual), the timed
# launches, then the fused reg_eval_reduce_kernel launches.
#   rd    : the L2's fabric-side read requests by size class -- exact bytes = 32 n32 + 64 n64 + 128 n128
#   fetch : rocprofv3's derived FETCH_SIZE (= RDREQ x 64 B on gfx950: half of a 128-B stream), kept
#           as the cross-check MI355X_MICROARCH.md "HBM" describes
#   write : derived WRITE_SIZE, calibrated on the known output bytes of the calibration dispatches
PMC_ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-solve --no-tsdf --calibrate"
REGEX="reg_eval_points|reg_eval_reduce"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum \
    --kernel-trace -f csv --kernel-include-regex "$REGEX" \
    -d $OUT/prof_rd -o rd -- python $REPO/bench.py $PMC_ARGS \
    > $OUT/prof_rd_bench.json 2> $OUT/prof_rd.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv --kernel-include-regex "$REGEX" \
    -d $OUT/prof_fetch -o fetch -- python $REPO/bench.py $PMC_ARGS \
    > $OUT/prof_fetch_bench.json 2> $OUT/prof_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv --kernel-include-regex "$REGEX" \
    -d $OUT/prof_write -o write -- python $REPO/bench.py $PMC_ARGS \
    > $OUT/prof_write_bench.json 2> $OUT/prof_write.err
# un-profiled full line (what the driver will see), N = 1
python $REPO/bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
cd $REPO
find gpurun_out/prof_stats gpurun_out/prof_rd gpurun_out/prof_fetch gpurun_out/prof_write -name '*.csv' | head -40
python profiles/summarize.py --round ${ROUND:-01} || true
