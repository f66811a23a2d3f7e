#!/bin/bash
# A/B of the sort-based TSDF paths' launch diet (round 4, second half): un-profiled ms per scan of
# profiles/merged_only.py (10-scan sessions of bench.py's two sensor shapes), same box, back to back:
#   BASE_DIR=<checkout>   a built checkout of the commit before the diet (git worktree add build/base_wt 40355f6;
#                         python -c "import __graft_entry__ as g; g.build()" in it), default build/base_wt if present:
#                         its own profiles/merged_only.py, library and bindings
#   VGX_DET_SWEEP=scan    the sweep as rocprim scan + det_seen_kernel (4 launches) instead of det_sweep_kernel (2)
#   VGX_TSDF_SORT=default rocprim's default radix_sort configuration instead of FewPassSort
#   VGX_DET_PHASES=1      host-side time between the points of a scan where the host waits
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
    env "$@" python $REPO/profiles/merged_only.py 2>&1 | grep "ms per scan" >> $OUT/tsdf_ab.txt
  done
}
BASE_DIR=${BASE_DIR:-$REPO/build/base_wt}
if [ -f $BASE_DIR/voxgraph_amd/lib/libvoxgraph_amd.so ]; then
  for cfg in "fast 1 reproducible" "merged 0 merged" "merged 1 merged+reproducible"; do
    set -- $cfg
    for rep in 1 2; do
      echo "--- $3: the commit before the diet (run $rep)" >> $OUT/tsdf_ab.txt
      (cd $BASE_DIR && INTEGRATOR=$1 DET=$2 python profiles/merged_only.py 2>&1 | grep "ms per scan") >> $OUT/tsdf_ab.txt
    done
  done
fi
run "reproducible: this tree"                 INTEGRATOR=fast DET=1
run "reproducible: scan sweeps"               INTEGRATOR=fast DET=1 VGX_DET_SWEEP=scan
run "reproducible: default sort"              INTEGRATOR=fast DET=1 VGX_TSDF_SORT=default
run "merged: this tree"                       INTEGRATOR=merged DET=0
run "merged: default sort"                    INTEGRATOR=merged DET=0 VGX_TSDF_SORT=default
run "merged, reproducible: this tree"         INTEGRATOR=merged DET=1
echo "--- phases, reproducible, this tree" >> $OUT/tsdf_ab.txt
for w in lidar rgbd; do
  VGX_DET_PHASES=1 INTEGRATOR=fast DET=1 WHICH=$w python $REPO/profiles/merged_only.py 2>&1 | grep -v amdgpu.ids >> $OUT/tsdf_ab.txt
done
