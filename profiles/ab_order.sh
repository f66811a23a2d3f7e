#!/bin/bash
# A/B of the XCD-aware launch order of the fused pass's tiles (VGX_FUSED_TILE_ORDER, vgx_reg.hip
# make_xcd_order): fused ms per solver evaluation on config 3 / full overlap / config 5, interleaved,
# two rounds; then one PMC pass per setting for the fused kernel's fabric read bytes.
#   gpurun -- 'bash profiles/ab_order.sh'
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --inner 2 --no-cpu-baseline --no-solve --no-tsdf --no-shipped --no-config2"
pick='import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1])
f,fo=d["fused"],d["roofline_full_overlap"]["fused"]
print("fused config3 %.3f ms | full overlap %.3f ms | config5 evaluation %.3f ms (solve %.0f ms)" % (
 f["ms_per_step"],fo["ms_per_step"],d["config5"]["registration_evaluation_ms"],d["config5"]["solve_ms"]))'
for round in 1 2; do
  for v in 0 1; do
    printf "round %s VGX_FUSED_TILE_ORDER=%s " $round $v
    VGX_FUSED_TILE_ORDER=$v timeout 300 python $REPO/bench.py --full-line $ARGS 2>$OUT/ab_order.err | python -c "$pick" || tail -3 $OUT/ab_order.err
  done
done
cd /tmp
for v in 0 1; do
  VGX_FUSED_TILE_ORDER=$v timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum \
      --kernel-trace -f csv --kernel-include-regex "reg_eval_reduce" -d $OUT/prof_order$v -o rd -- \
      python $REPO/bench.py --steps 3 --inner 1 --warmup 1 --no-cpu-baseline --no-solve --no-tsdf --no-shipped --no-config5 --no-config2 \
      > /dev/null 2> $OUT/prof_order$v.err
  python - <<PY
import csv, glob, collections
d = collections.OrderedDict()
for f in glob.glob("$OUT/prof_order$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        e = d.setdefault((int(r["Grid_Size"]), int(r["Dispatch_Id"])), {})
        e[r["Counter_Name"]] = float(r["Counter_Value"])
g = collections.OrderedDict()
for (grid, _), c in d.items():
    g.setdefault(grid, []).append(32 * c.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * c.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * c.get("TCC_EA0_RDREQ_128B_sum", 0))
print("VGX_FUSED_TILE_ORDER=$v fused read GB per launch by workload (grid):", {k: round(sum(v) / len(v) / 1e9, 3) for k, v in g.items()})
PY
done
