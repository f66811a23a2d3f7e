#!/bin/bash
# racing TSDF kernel on the config-2 city session (long rays, 60 % of the points cast a ray) next to the room sessions:
#   /usr/local/graft/bin/gpurun --timeout 900 -- bash profiles/probes/run_city.sh
OUT=gpurun_out/r05_tsdf; mkdir -p $OUT
VGX_TSDF_KERNEL=v1 timeout 200 python profiles/probes/tsdf_city_probe.py 2>/dev/null | tail -1
timeout 200 python profiles/probes/tsdf_city_probe.py 2>/dev/null | tail -1
VGX_PROBE_ORGANISED=1 timeout 200 python profiles/probes/tsdf_city_probe.py 2>/dev/null | tail -1
timeout 100 python profiles/probes/tsdf_racing_probe.py > $OUT/coop.json 2>/dev/null; python -c "
import json; j=json.load(open('gpurun_out/r05_tsdf/coop.json'))
for k,v in j.items():
    if isinstance(v,dict): print(k[:5], 'median', round(v['kernel_us_median'],1), 'b2b', round(v['back_to_back_us'],1))"
