#!/bin/bash
# The shipped sampling configuration by itself under rocprofv3 --kernel-trace --stats: per-kernel time of the
# mt19937 streams, the draws and the fused kernel (default = quad bricks on demand, then apron for comparison).
#   gpurun -- 'bash profiles/shipped_only.sh'
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 20 --warmup 1 --inner 1 --no-cpu-baseline --no-tsdf --no-solve --no-config5 --no-config2 --no-multi-ctx --no-parity --no-full-overlap"
rm -rf $OUT/prof_shipped
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_shipped -o shipped -- \
    python $REPO/bench.py $ARGS --detail $OUT/shipped_detail.json > $OUT/shipped_line.json 2> $OUT/shipped.err
python - <<PY
import csv, glob, json
d = json.load(open("$OUT/shipped_detail.json"))
s = d["shipped_config"]
print(json.dumps({k: s.get(k) for k in ("ms_per_evaluation", "stream_ms_per_evaluation", "residuals_per_evaluation", "brick_layout_chosen")}))
print("apron:", json.dumps({k: s["apron_bricks"].get(k) for k in ("ms_per_evaluation", "stream_ms_per_evaluation", "cost_equals_default")}) if s.get("apron_bricks") else None)
print("fused config3:", d["fused"]["ms_per_step"], "points kernel_ms:", d["roofline"]["kernel_ms"])
for f in glob.glob("$OUT/prof_shipped/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), "avg us", round(float(r["AverageNs"]) / 1e3, 1), "total ms", round(float(r["TotalDurationNs"]) / 1e6, 2))
PY
