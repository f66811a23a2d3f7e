#!/bin/bash
# HBM byte counters for the fused normal-equation kernel (reg_eval_reduce_kernel), one
# --pmc pass per counter, same corrections as collect.sh/summarize.py (FETCH_SIZE x1.997,
# WRITE_SIZE x0.998 for plain stores).  Output: gpurun_out/prof_fused_{fetch,write}.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-solve --no-tsdf"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -f csv --kernel-include-regex reg_eval_reduce \
      -d $OUT/prof_fused_$c -o pmc -- python $REPO/bench.py $ARGS \
      > $OUT/prof_fused_$c.json 2> $OUT/prof_fused_$c.err
done
cd $REPO
python - <<'PY'
import csv, glob
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/prof_fused_{c}/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c]
        vals = [float(r["Counter_Value"]) for r in rows]
        print(c, len(vals), "dispatches; KiB per dispatch:", [round(v) for v in vals[:8]])
PY
