#!/bin/bash
# A/B of the XCD-aware launch order of the materialising pass's tiles (VGX_POINTS_TILE_ORDER,
# vgx_reg.hip make_xcd_order, uniform_work): ms per pass on config 3 / full overlap, interleaved, two
# rounds, crossed with the chunk culling of that pass (VGX_POINTS_CULL); then one PMC pass per setting for the points kernel's fabric read bytes.
#   gpurun -- 'bash profiles/ab_porder.sh'
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --inner 4 --no-cpu-baseline --no-solve --no-tsdf --no-shipped --no-config2 --no-config5 --no-fused --no-multi-ctx"
pick='import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1])
fo=d["roofline_full_overlap"]
print("points kernel config3 %.3f ms/pass (%.1f G evals/s) | full overlap %.3f ms (%.1f G evals/s)" % (
 d["roofline"]["kernel_ms"],d["value"]/1e3,fo["kernel_ms"],fo["value"]/1e3))'
for round in 1 2; do
  for v in 0 1; do for k in 0 1; do
    printf "round %s VGX_POINTS_TILE_ORDER=%s VGX_POINTS_CULL=%s " $round $v $k
    VGX_POINTS_CULL=$k VGX_POINTS_TILE_ORDER=$v timeout 300 python $REPO/bench.py --full-line $ARGS 2>$OUT/ab_porder.err | python -c "$pick" || tail -3 $OUT/ab_porder.err
  done; done
done
cd /tmp
for v in 0 1; do
  VGX_POINTS_TILE_ORDER=$v timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum \
      --kernel-trace -f csv --kernel-include-regex "reg_eval_points" -d $OUT/prof_porder$v -o rd -- \
      python $REPO/bench.py --steps 3 --inner 1 --warmup 1 --no-cpu-baseline --no-solve --no-tsdf --no-shipped --no-config5 --no-config2 \
      > /dev/null 2> $OUT/prof_porder$v.err
  python - <<PY
import csv, glob, collections
d = collections.OrderedDict()
for f in glob.glob("$OUT/prof_porder$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        e = d.setdefault((int(r["Grid_Size"]), int(r["Dispatch_Id"])), {})
        e[r["Counter_Name"]] = float(r["Counter_Value"])
g = collections.OrderedDict()
for (grid, _), c in d.items():
    g.setdefault(grid, []).append(32 * c.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * c.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * c.get("TCC_EA0_RDREQ_128B_sum", 0))
print("VGX_POINTS_TILE_ORDER=$v points-kernel read GB per launch by workload (grid):", {k: round(sum(v) / len(v) / 1e9, 3) for k, v in g.items()})
PY
done
