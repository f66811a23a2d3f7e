#!/bin/bash
# Runs the whole GPU suite N times the way the driver runs it (-x: stop at the first failure) and
# records one line per run:   gpurun -- 'N=5 bash profiles/suite_runs.sh'
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
: > $OUT/suite_runs.txt
for k in $(seq 1 ${N:-5}); do
  python -m pytest $REPO/tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -1 | sed "s/^/run $k: /" | tee -a $OUT/suite_runs.txt
done
