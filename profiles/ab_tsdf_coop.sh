#!/bin/bash
# racing TSDF kernel, round 5: the cooperative kernel (vgx_tsdf_coop.hip) against the one-thread-per-point kernel
# (VGX_TSDF_KERNEL=v1) on one lease; the TSDF test files first (parity before speed)
OUT=gpurun_out/r05_tsdf
mkdir -p $OUT
timeout 900 python -m pytest tests/test_tsdf_gpu.py tests/test_tsdf_dropin_gpu.py tests/test_errors_gpu.py -x -q -m gpu > $OUT/pytest_tsdf.txt 2>&1
tail -5 $OUT/pytest_tsdf.txt
for i in 1 2; do
  VGX_TSDF_KERNEL=v1 timeout 300 python profiles/probes/tsdf_racing_probe.py > $OUT/probe_v1_$i.json 2> $OUT/probe_v1_$i.err
  timeout 300 python profiles/probes/tsdf_racing_probe.py > $OUT/probe_v2_$i.json 2> $OUT/probe_v2_$i.err
done
python - <<'PY'
import json
for n in ("v1_1", "v2_1", "v1_2", "v2_2"):
    try:
        j = json.load(open(f"gpurun_out/r05_tsdf/probe_{n}.json"))
    except Exception as e:
        print(n, "failed", e); continue
    for k, v in j.items():
        if isinstance(v, dict):
            print(n, k[:5], "kernel us %.1f (min %.1f max %.1f first %.1f) b2b %.1f" % (v["kernel_us"], v["kernel_us_min"], v["kernel_us_max"], v["kernel_us_first_scan"], v["back_to_back_us"]), v["per_scan"])
PY
