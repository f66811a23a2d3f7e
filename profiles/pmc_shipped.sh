#!/bin/bash
# HBM read traffic of the sampling path's kernels (mt19937 streams, draws, point gather, fused kernel) on the
# shipped configuration: one --pmc pass (TCC_EA0_RDREQ size classes), never combined with hip/hsa traces.
#   gpurun -- 'bash profiles/pmc_shipped.sh'
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 3 --warmup 1 --inner 1 --no-cpu-baseline --no-tsdf --no-solve --no-config5 --no-config2 --no-multi-ctx --no-parity --no-full-overlap ${EXTRA_ARGS:-}"
rm -rf $OUT/prof_shipped_rd $OUT/prof_shipped_wr
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum \
    --kernel-trace -f csv --kernel-include-regex "reg_draw|reg_gather|mt_generate|reg_eval_reduce" \
    -d $OUT/prof_shipped_rd -o rd -- python $REPO/bench.py $ARGS --detail $OUT/shipped_rd_detail.json > /dev/null 2> $OUT/shipped_rd.err
timeout 300 rocprofv3 --pmc WRITE_SIZE \
    --kernel-trace -f csv --kernel-include-regex "reg_draw|reg_gather|mt_generate|reg_eval_reduce" \
    -d $OUT/prof_shipped_wr -o wr -- python $REPO/bench.py $ARGS --detail $OUT/shipped_wr_detail.json > /dev/null 2> $OUT/shipped_wr.err
python - <<PY
import csv, glob, collections
def table(pattern):
    d = collections.OrderedDict()
    for f in glob.glob(pattern, recursive=True):
        for x in csv.DictReader(open(f)):
            k = (x["Kernel_Name"].split("(")[0][-48:], int(x["Grid_Size"]))
            e = d.setdefault(k, collections.defaultdict(list))
            e[x["Counter_Name"]].append(float(x["Counter_Value"]))
    return d
rd = table("$OUT/prof_shipped_rd/**/*counter_collection.csv")
wr = table("$OUT/prof_shipped_wr/**/*counter_collection.csv")
for k, e in rd.items():
    n = len(e["TCC_EA0_RDREQ_sum"])
    rb = (32 * sum(e["TCC_EA0_RDREQ_32B_sum"]) + 64 * sum(e["TCC_EA0_RDREQ_64B_sum"]) + 128 * sum(e["TCC_EA0_RDREQ_128B_sum"])) / n
    w = wr.get(k, {}).get("WRITE_SIZE", [])
    wb = sum(w) / len(w) * 1024 if w else float("nan")        # WRITE_SIZE is in KB on gfx950 (MI355X_MICROARCH.md)
    print(k[0].ljust(48), str(k[1]).rjust(9), "launches", n, "read MB %.1f  write MB (uncalibrated) %.1f" % (rb / 1e6, wb / 1e6))
PY
