#!/bin/bash
# racing TSDF kernel: 128 instead of 256 points per workgroup for scans up to VGX_TSDF_SPLIT_BELOW points (twice the lanes per
# ray, two wavefronts per SIMD on a 64 x 1024 LiDAR scan); session-old integrator (tsdf_phase_probe.py), fresh (tsdf_racing_probe.py)
OUT=gpurun_out/tsdf_split
mkdir -p $OUT
for rep in 1 2; do
for c in 0 100000 400000; do
  echo "=== VGX_TSDF_SPLIT_BELOW=$c (rep $rep)"
  VGX_TSDF_SPLIT_BELOW=$c timeout 200 python profiles/probes/tsdf_phase_probe.py 2> $OUT/phase_$c.err | cut -c1-420
  VGX_TSDF_SPLIT_BELOW=$c timeout 200 python profiles/probes/tsdf_racing_probe.py > $OUT/racing_$c.json 2> $OUT/racing_$c.err
  python - <<PY
import json
j = json.load(open("$OUT/racing_$c.json"))
for k, v in j.items():
    if isinstance(v, dict):
        print("   fresh", k[:5], "kernel median %.1f us, back to back %.1f us" % (v["kernel_us_median"], v["back_to_back_us"]), "trace", {a: round(b, 1) for a, b in (v["trace"] or {}).items() if a in ("wg_us_mean", "wg_us_max", "rays_max", "rounds_max", "folds_max", "span_us")})
PY
done
done 2>&1 | tee $OUT/summary.txt
