#!/bin/bash
# A/B of the grouped fused kernel (VGX_FUSED_GROUPS=1: a thread evaluates its points against up to
# VGX_FUSED_GROUP_M constraints that share the reference submap, reg_eval_reduce_group_kernel) against
# the lean kernel for everything (=0): fused ms per solver evaluation on config 3 / full overlap /
# config 5, interleaved; then one PMC pass per setting for the fused kernels' fabric read bytes.
#   gpurun -- 'bash profiles/ab_group.sh'
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --inner 2 --no-cpu-baseline --no-solve --no-tsdf --no-shipped --no-config2"
pick='import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1])
f,fo=d["fused"],d["roofline_full_overlap"]["fused"]
print("fused config3 %.3f ms (rel.err %.1e) | full overlap %.3f ms | config5 evaluation %.3f ms" % (
 f["ms_per_step"],f["cost_vs_materialised"],fo["ms_per_step"],d["config5"]["registration_evaluation_ms"]))'
for cfg in "0 2" "1 2" "1 3" "1 4" "0 2" "1 2"; do
  set -- $cfg
  printf "VGX_FUSED_GROUPS=%s M=%s " $1 $2
  VGX_FUSED_GROUPS=$1 VGX_FUSED_GROUP_M=$2 timeout 300 python $REPO/bench.py $ARGS 2>$OUT/ab_group.err | python -c "$pick" || tail -3 $OUT/ab_group.err
done
cd /tmp
for cfg in "0 2" "1 2" "1 4"; do
  set -- $cfg
  VGX_FUSED_GROUPS=$1 VGX_FUSED_GROUP_M=$2 timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum \
      --kernel-trace -f csv --kernel-include-regex "reg_eval_reduce" -d $OUT/prof_group$1$2 -o rd -- \
      python $REPO/bench.py --steps 3 --inner 1 --warmup 1 --no-cpu-baseline --no-solve --no-tsdf --no-shipped --no-config5 --no-config2 \
      > /dev/null 2> $OUT/prof_group$1$2.err
  python - <<PY
import csv, glob, collections
d = collections.OrderedDict()
for f in glob.glob("$OUT/prof_group$1$2/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        e = d.setdefault((int(r["Grid_Size"]), int(r["Dispatch_Id"])), {})
        e[r["Counter_Name"]] = float(r["Counter_Value"])
g = collections.OrderedDict()
for (grid, _), c in d.items():
    g.setdefault(grid, []).append(32 * c.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * c.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * c.get("TCC_EA0_RDREQ_128B_sum", 0))
print("VGX_FUSED_GROUPS=$1 M=$2 fused read GB per launch by kernel grid:", {k: round(sum(v) / len(v) / 1e9, 3) for k, v in g.items()})
PY
done
