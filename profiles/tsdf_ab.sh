#!/bin/bash
# A/B of the sort-based TSDF paths' launch diet (round 4): un-profiled ms per scan of profiles/merged_only.py with
#   VGX_DET_SWEEP=scan    the sweep as rocprim scan + det_seen_kernel (4 launches) instead of det_sweep_kernel (2)
#   VGX_TSDF_SORT=default rocprim's default radix_sort config instead of FewPassSort
#   gpurun -- 'bash profiles/tsdf_ab.sh'  -> gpurun_out/tsdf_ab.txt
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
: > $OUT/tsdf_ab.txt
run() {  # label, env...
  local label=$1; shift
  for rep in 1 2; do
    echo "--- $label (run $rep)" >> $OUT/tsdf_ab.txt
    env "$@" python $REPO/profiles/merged_only.py >> $OUT/tsdf_ab.txt 2>&1
  done
}
run "fast det: new"                   INTEGRATOR=fast DET=1
run "fast det: scan sweeps"           INTEGRATOR=fast DET=1 VGX_DET_SWEEP=scan
run "fast det: default sort"          INTEGRATOR=fast DET=1 VGX_TSDF_SORT=default
run "fast det: scan + default sort"   INTEGRATOR=fast DET=1 VGX_DET_SWEEP=scan VGX_TSDF_SORT=default
run "merged: new"                     INTEGRATOR=merged DET=0
run "merged: default sort"            INTEGRATOR=merged DET=0 VGX_TSDF_SORT=default
run "merged det: new"                 INTEGRATOR=merged DET=1
