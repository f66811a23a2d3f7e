#!/bin/bash
# reproducible / merged TSDF modes: the counters' read-back through pinned report words (default) against the copy +
# stream synchronisation of rounds 3-4 (VGX_DET_REPORT=0), alternating processes on one lease
#   /usr/local/graft/bin/gpurun --timeout 900 -- bash profiles/probes/run_det_report.sh
for k in 1 2; do
  for r in 0 1; do
    VGX_DET_REPORT=$r timeout 300 python profiles/probes/tsdf_age_probe.py 2>/dev/null > gpurun_out/det_report_$r.json
    python - <<PY
import json
d = json.load(open("gpurun_out/det_report_$r.json"))
for k, v in d.items():
    rep = [x["back_to_back_us"] for x in v["reproducible"]]
    mer = [x["back_to_back_us"] for x in v["merged"]]
    print("VGX_DET_REPORT=$r", k[:5], "reproducible us per scan, fresh %.1f, session-old %.1f (median of the last six passes); merged %.1f" % (rep[0], sorted(rep[-6:])[3], sorted(mer)[1]))
PY
  done
done
