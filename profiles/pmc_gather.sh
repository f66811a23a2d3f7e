#!/bin/bash
# L1 / texture-address path counters of the fused kernel (config 3 and full-overlap grids), one small
# --pmc pass per group, never combined with API traces (TA_TA_BUSY_sum / TA_FLAT_READ_WAVEFRONTS_sum /
# TA_ADDR_STALLED_* abort rocprofv3 on this image: left out):  gpurun -- 'bash profiles/pmc_gather.sh'
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(TA|TCP|TD|GRBM|SQ|SQC|TCC)_[A-Za-z0-9_]+" | sort -u > $OUT/pmc_names.txt
ARGS="--steps 2 --warmup 1 --inner 1 --no-cpu-baseline --no-solve --no-tsdf --no-shipped --no-config5 --no-config2 --no-multi-ctx --no-parity --no-fo-plain"
i=0
while read -r group; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $group --kernel-trace -f csv --kernel-include-regex "reg_eval_reduce" \
      -d $OUT/prof_ta_$i -o ta -- python $REPO/bench.py $ARGS > /dev/null 2> $OUT/prof_ta_$i.err \
      || echo "group $i ($group) failed: $(tail -2 $OUT/prof_ta_$i.err | tr '\n' ' ')"
done <<'GROUPS'
GRBM_GUI_ACTIVE TA_BUSY_avr TA_BUSY_max
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum
TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum
TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum
GROUPS
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/prof_ta_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:40], r["Grid_Size"], r["Counter_Name"])
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
        for k, (s, n) in sorted(acc.items()):
            print(k[1], k[2], "%.4g per dispatch (%d)" % (s / n, n))
PY
