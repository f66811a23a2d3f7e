#!/bin/bash
# racing TSDF kernel: which points a workgroup takes from an unorganised cloud (VGX_TSDF_CHUNK = log2 of the run length;
# 0 = 256 consecutive points), session-old integrator (tsdf_phase_probe.py) and fresh one (tsdf_racing_probe.py)
#   /usr/local/graft/bin/gpurun --timeout 900 -- bash profiles/probes/run_tsdf_chunk.sh
OUT=gpurun_out/tsdf_chunk
mkdir -p $OUT
for rep in 1; do
for c in ${CHUNKS:-0 4 5}; do
  echo "=== VGX_TSDF_CHUNK=$c (rep $rep)"
  VGX_TSDF_CHUNK=$c timeout 200 python profiles/probes/tsdf_phase_probe.py 2> $OUT/phase_$c.err
  VGX_TSDF_CHUNK=$c timeout 200 python profiles/probes/tsdf_racing_probe.py > $OUT/racing_$c.json 2> $OUT/racing_$c.err
  python - <<PY
import json
j = json.load(open("$OUT/racing_$c.json"))
for k, v in j.items():
    if isinstance(v, dict):
        print("   fresh", k[:5], "kernel median %.1f us, back to back %.1f us, one-point %.1f" % (v["kernel_us_median"], v["back_to_back_us"], v["one_point_scan_us"]), "trace", {a: round(b, 1) for a, b in (v["trace"] or {}).items() if a in ("wg_us_mean", "wg_us_max", "rays_max", "rounds_max", "folds_max", "span_us")})
PY
done
done 2>&1 | tee $OUT/summary.txt
