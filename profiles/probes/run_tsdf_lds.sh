#!/bin/bash
# racing TSDF kernel with the 52-byte ray record (40.7 KB of LDS: four workgroups per CU) against the commit before (46.6 KB:
# three), libraries alternated on one lease: tsdf_phase_probe.py (session-old integrator) and tsdf_racing_probe.py (fresh)
for rep in 1 2; do
for lib in libvoxgraph_amd.so libvoxgraph_amd_old.so; do
  echo "=== $lib (rep $rep)"
  VGX_LIB=$PWD/voxgraph_amd/lib/$lib timeout 200 python profiles/probes/tsdf_phase_probe.py 2>/dev/null | cut -c1-330
  VGX_LIB=$PWD/voxgraph_amd/lib/$lib timeout 200 python profiles/probes/tsdf_racing_probe.py > /tmp/racing.json 2>/dev/null
  python - <<PY
import json
j = json.load(open("/tmp/racing.json"))
for k, v in j.items():
    if isinstance(v, dict):
        print("   fresh", k[:5], "kernel median %.1f us, back to back %.1f us" % (v["kernel_us_median"], v["back_to_back_us"]))
PY
done
done
