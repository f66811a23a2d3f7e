#!/bin/bash
# which counter separates a SLOW placement of the materialising pass's output arrays from a FAST one?
# Every pass below is its own process (rocprofv3 --pmc, <= 4 TCC counters per pass, never combined with hip / hsa / sys
# traces): profiles/probes/points_placement_pmc.py finds a slow and a fast set IN that process and ends with labelled
# launches alternating between the two; profiles/placement_pmc_summary.py lines the counter rows up with the labels.
# Run through gpurun from the repo root; outputs in gpurun_out/placement_pmc/.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/placement_pmc
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run_pass() {   # name, counters...
  local name=$1; shift
  VGX_PROBE_OUT=$OUT/${name}_labels.json timeout 300 rocprofv3 --pmc "$@" --kernel-trace -f csv json \
      --kernel-include-regex "reg_eval_points_kernel" -d $OUT/$name -o $name -- python $REPO/profiles/probes/points_placement_pmc.py \
      > $OUT/${name}.out 2> $OUT/${name}.err
  echo "pass $name rc=$?"
}
# un-profiled first: the two speeds exist without a profiler attached
VGX_PROBE_OUT=$OUT/unprofiled_labels.json python $REPO/profiles/probes/points_placement_pmc.py > $OUT/unprofiled.out 2> $OUT/unprofiled.err
run_pass wr_a   TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WR_UNCACHED_32B_sum
run_pass wr_b   TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum
run_pass rd_a   TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_SRC_FIFO_FULL_sum
run_pass tcc_c  TCC_BUSY_sum TCC_CYCLE_sum TCC_LATENCY_FIFO_FULL_sum TCC_IB_STALL_sum
run_pass tcp_a  TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
run_pass tcp_b  TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_TCC_READ_REQ_sum
# per channel: the raw counters keep their dimensions (16 TCC instances x 8 XCC) in the json output
run_pass ch_wr  TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL
run_pass ch_lvl TCC_EA0_WRREQ_LEVEL TCC_EA0_WRREQ_DRAM_CREDIT_STALL
cd $REPO
python profiles/placement_pmc_summary.py $OUT > $OUT/summary.txt 2>&1
tail -60 $OUT/summary.txt
