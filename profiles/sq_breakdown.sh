#!/bin/bash
# Where the wave cycles of the two REG kernels and of the TSDF kernel go (SQ counters, one pass; units = quad-cycles per
# MI355X_MICROARCH.md): ACTIVE_INST_* = issuing, WAIT_ANY = parked on s_waitcnt/barrier,
# WAIT_INST_ANY = issue stall.  WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES \
    --kernel-trace -f csv --kernel-include-regex "reg_eval_points_kernel|reg_eval_reduce|tsdf_integrate" \
    -d $OUT/prof_sq -o sq -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-solve \
    > /dev/null 2> $OUT/prof_sq.err
cd $REPO
python - <<'PY'
import csv, glob, json, collections
d = collections.OrderedDict()
for f in glob.glob("gpurun_out/prof_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "fused" if "reduce" in r["Kernel_Name"] else ("tsdf_integrate" if "tsdf" in r["Kernel_Name"] else "materialising")
        d.setdefault(k, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in d.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0) or 1
    out[k] = {"dispatches": len(next(iter(c.values()))), **{n: v for n, v in m.items()},
              "frac_wait_any": m.get("SQ_WAIT_ANY", 0) / wc, "frac_wait_inst_any": m.get("SQ_WAIT_INST_ANY", 0) / wc,
              "frac_active_inst_any": m.get("SQ_ACTIVE_INST_ANY", 0) / wc,
              "frac_active_inst_valu": m.get("SQ_ACTIVE_INST_VALU", 0) / wc,
              "valu_insts_per_wave": m.get("SQ_INSTS_VALU", 0) / (m.get("SQ_WAVES", 0) or 1),
              "salu_insts_per_wave": m.get("SQ_INSTS_SALU", 0) / (m.get("SQ_WAVES", 0) or 1)}
json.dump(out, open("gpurun_out/sq_breakdown.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
