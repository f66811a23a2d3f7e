#!/bin/bash
# Launch-by-launch view of ONE scan of the sort-based TSDF paths (profiles/launch_sequence.py):
#   gpurun -- 'bash profiles/tsdf_launches.sh'   -> gpurun_out/tsdf_launches.txt
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
: > $OUT/tsdf_launches.txt
# the sources this trace describes (harness/bench_tsdf.py profiled_launches drops the counts when they differ)
echo "sources sha256: $(cd $REPO && python -c 'from harness.bench_tsdf import tsdf_sources_sha; print(tsdf_sources_sha())')" >> $OUT/tsdf_launches.txt
for cfg in "fast 1 lidar det_points" "fast 1 rgbd det_points" "merged 0 lidar merged_bundle" "merged 0 rgbd merged_bundle"; do
  set -- $cfg
  rm -rf $OUT/prof_tl
  INTEGRATOR=$1 DET=$2 WHICH=$3 timeout 300 rocprofv3 --kernel-trace -f csv -d $OUT/prof_tl -o t -- python $REPO/profiles/merged_only.py >> $OUT/tsdf_launches.txt 2>/dev/null
  echo "=== $1 det=$2 $3" >> $OUT/tsdf_launches.txt
  python $REPO/profiles/launch_sequence.py $OUT/prof_tl $4 >> $OUT/tsdf_launches.txt 2>&1
done
# un-profiled timings
for cfg in "fast 1" "merged 0" "merged 1"; do
  set -- $cfg
  INTEGRATOR=$1 DET=$2 python $REPO/profiles/merged_only.py >> $OUT/tsdf_launches.txt 2>/dev/null
done
