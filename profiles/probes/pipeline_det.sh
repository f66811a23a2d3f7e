#!/bin/bash
# The config-2 stand-in session (city LiDAR: 16 m rays, 80 voxels long) in the reproducible TSDF mode under
# different speculation settings: ms per scan, from bench.py's pipeline block.
#   gpurun -- 'bash profiles/probes/pipeline_det.sh'
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
run() {
  echo "== $*"
  env "$@" python3 $REPO/bench.py --gpus 1 --steps 2 --warmup 1 --no-tsdf --no-config5 --no-solve --no-cpu-baseline --no-shipped \
      --no-parity --no-full-overlap --no-multi-ctx --detail $REPO/gpurun_out/pd_detail.json > /dev/null 2> $REPO/gpurun_out/pd_err.txt
  python3 - <<PY
import json
p = json.load(open("$REPO/gpurun_out/pd_detail.json"))["pipeline_config2"]
print("  racing %.3f ms/scan   reproducible %.3f ms/scan   rmse %.4f / %.4f" % (p["tsdf_integrate_ms_per_scan"],
      p["reproducible_tsdf_mode"]["tsdf_integrate_ms_per_scan"], p["xy_rmse_m_optimised"], p["reproducible_tsdf_mode"]["xy_rmse_m_optimised"]))
PY
}
for s in ${SETTINGS:-"X=0" "VGX_DET_CAP=8" "VGX_DET_CAP=16" "VGX_DET_CAP=8 VGX_DET_MARK_LIFE=4" "VGX_DET_CAP_THRESHOLD=16000000"}; do run $s; done
