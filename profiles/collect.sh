#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's `roofline`, `roofline_full_overlap`, `fused` and
# `tsdf.*.roofline` objects on the GPU box (run through gpurun from the repo root):
#   stats : rocprofv3 --kernel-trace --stats of the SAME command the driver runs; per-dispatch
#           durations are grouped per (kernel, grid size), i.e. per workload, by summarize.py
#   rd / fetch / write : HBM byte counters, one --pmc pass each (FETCH_SIZE takes 3 TCC slots,
#           WRITE_SIZE 2: MI355X_MICROARCH.md "rocprofv3 PMC slots"), never combined with
#           hip/hsa/sys traces.  The command launches, in order: two calibration dispatches of
#           reg_eval_points_kernel (poses 10 km apart: no evaluation finds a reading block, so each
#           writes exactly 36 B per residual and -- every tile being culled -- reads no points),
#           config 3's launches, the full-overlap workload's (--no-fo-plain: in the shipped launch
#           order only, so that a grid size names one workload), then the fused kernel on both.
#   rd_plain / write_plain : the same with VGX_POINTS_TILE_ORDER=0 (plain constraint-major order)
#   sq    : SQ wave-cycle breakdown of the REG kernels and the TSDF kernel
# Outputs land in gpurun_out/prof_*; profiles/summarize.py turns them into profiles/rNN_*.json and
# profiles/hbm_traffic.json.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_stats -o stats -- \
    python $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --detail $OUT/prof_stats_detail.json > $OUT/prof_stats_bench.json 2> $OUT/prof_stats.err
# (--placement-candidates 1: the counter passes keep the launch sequence summarize.py indexes -- no placement trials in front)
PMC_ARGS="--placement-candidates 1 --steps 3 --inner 1 --warmup 1 --no-cpu-baseline --no-solve --no-tsdf --no-config5 --no-config2 --no-fo-plain --no-multi-ctx --no-parity --calibrate"
REGEX="reg_eval_points|reg_eval_reduce"
timeout 240 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum \
    --kernel-trace -f csv --kernel-include-regex "$REGEX" \
    -d $OUT/prof_rd -o rd -- python $REPO/bench.py $PMC_ARGS --detail $OUT/prof_rd_detail.json \
    > $OUT/prof_rd_bench.json 2> $OUT/prof_rd.err
timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv --kernel-include-regex "$REGEX" \
    -d $OUT/prof_fetch -o fetch -- python $REPO/bench.py $PMC_ARGS --detail $OUT/prof_fetch_detail.json \
    > $OUT/prof_fetch_bench.json 2> $OUT/prof_fetch.err
timeout 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv --kernel-include-regex "$REGEX" \
    -d $OUT/prof_write -o write -- python $REPO/bench.py $PMC_ARGS --detail $OUT/prof_write_detail.json \
    > $OUT/prof_write_bench.json 2> $OUT/prof_write.err
# the full-overlap workload in plain constraint-major launch order (bench.py's plain_order object)
for c in rd write; do
  if [ $c = rd ]; then CNT="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"; else CNT="WRITE_SIZE"; fi
  VGX_POINTS_TILE_ORDER=0 timeout 300 rocprofv3 --pmc $CNT --kernel-trace -f csv --kernel-include-regex "reg_eval_points" \
      -d $OUT/prof_${c}_plain -o ${c}_plain -- python $REPO/bench.py $PMC_ARGS --no-fused --no-shipped --detail $OUT/prof_${c}_plain_detail.json \
      > $OUT/prof_${c}_plain_bench.json 2> $OUT/prof_${c}_plain.err
done
timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES \
    --kernel-trace -f csv --kernel-include-regex "reg_eval_points_kernel|reg_eval_reduce|tsdf_integrate|det_apply|det_sweep|det_ray" \
    -d $OUT/prof_sq -o sq -- python $REPO/bench.py --placement-candidates 1 --steps 2 --inner 1 --warmup 1 --no-cpu-baseline --no-solve --no-shipped --no-config5 --no-config2 --no-multi-ctx --no-parity \
    > /dev/null 2> $OUT/prof_sq.err
# un-profiled, the driver's EXACT command from the repo root (what BENCH_rNN.json will hold): stdout = the one
# compact line (harness/bench_line.py), the full object in bench_detail.json next to bench.py
# gpurun brings at most 64 MiB of gpurun_out/ back: the traces compress 15-20 x (summarize.py reads .csv.gz), and the
# kernel traces of the counter passes (summarize.py reads their counter_collection only) go
find $OUT/prof_* -name "*.csv" -size +1M -exec gzip -9 {} \;
find $OUT/prof_rd $OUT/prof_fetch $OUT/prof_write $OUT/prof_rd_plain $OUT/prof_write_plain $OUT/prof_sq -name "*kernel_trace.csv*" -delete 2>/dev/null
du -sh $OUT
cd $REPO
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_cmd.out 2> $OUT/driver_cmd.err
cp bench_detail.json $OUT/bench_full.json
python profiles/summarize.py --round ${ROUND:-04} || true
