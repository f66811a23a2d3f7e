#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's `roofline`, `roofline_full_overlap`, `fused` and
# `tsdf.*.roofline` objects on the GPU box (run through gpurun from the repo root):
#   stats : rocprofv3 --kernel-trace --stats of the SAME command the driver runs; per-dispatch
#           durations are grouped per (kernel, grid size), i.e. per workload, by summarize.py
#   rd / fetch / write : HBM byte counters, one --pmc pass each (FETCH_SIZE takes 3 TCC slots,
#           WRITE_SIZE 2: MI355X_MICROARCH.md "rocprofv3 PMC slots"), never combined with
#           hip/hsa/sys traces.  The command launches, in order: two calibration dispatches of
#           reg_eval_points_kernel (poses 10 km apart: no evaluation finds a reading block, so each
#           writes exactly 36 B per residual), config 3's launches, the full-overlap workload's, then
#           the fused kernel on both workloads.
#   sq    : SQ wave-cycle breakdown of the REG kernels and the TSDF kernel
# Outputs land in gpurun_out/prof_*; profiles/summarize.py turns them into profiles/rNN_*.json and
# profiles/hbm_traffic.json.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_stats -o stats -- \
    python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/prof_stats_bench.json 2> $OUT/prof_stats.err
PMC_ARGS="--steps 3 --inner 1 --warmup 1 --no-cpu-baseline --no-solve --no-tsdf --no-shipped --no-config5 --no-config2 --calibrate"
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
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES \
    --kernel-trace -f csv --kernel-include-regex "reg_eval_points_kernel|reg_eval_reduce|tsdf_integrate" \
    -d $OUT/prof_sq -o sq -- python $REPO/bench.py --steps 2 --inner 1 --warmup 1 --no-cpu-baseline --no-solve --no-shipped --no-config5 --no-config2 \
    > /dev/null 2> $OUT/prof_sq.err
# un-profiled full line (what the driver will see), N = 1
python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_full.json 2> $OUT/bench_full.err
cd $REPO
python profiles/summarize.py --round ${ROUND:-02} || true
