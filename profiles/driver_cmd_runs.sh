#!/bin/bash
# the driver's exact command N times in a row on one lease: how the headline moves from process to process (the placement
# lottery: every process gets other physical pages for its candidate arrays)   gpurun -- 'N=5 bash profiles/driver_cmd_runs.sh'
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
mkdir -p gpurun_out
: > gpurun_out/driver_cmd_runs.txt
for k in $(seq 1 ${N:-5}); do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > /tmp/line.json 2> /tmp/line.err
  python3 - "$k" <<'PY' | tee -a gpurun_out/driver_cmd_runs.txt
import json, sys
l = [x for x in open("/tmp/line.json") if x.startswith("{")][-1]
d = json.loads(l)
r, f, t = d["roofline"], d["fused"], d["tsdf"]
lid = [v for k, v in t.items() if k.startswith("lidar")][0]
rgb = [v for k, v in t.items() if k.startswith("rgbd")][0]
print("run %s: value %.1f G/s  %.3f ms/pass  frac %.3f  first %.3f  median %.3f  blocked %.3f  f64 rows %.2f ms | fused %.3f ms  cost only %.3f | solve %.1f ms | "
      "tsdf lidar %.4f  rgbd %.4f ms/scan  latency under solve p50/p99 %d/%d  %d/%d us  over 1 ms %s/%s | line %d B" % (
          sys.argv[1], d["value"] / 1e3, d["ms_per_step"] / 25, r["frac"], r["frac_first_allocation"], r["frac_median_allocation"],
          r["frac_blocked_layout"], r["f64_rows_ms"], f["stream_ms_per_step"], f["cost_only_ms"], d["solve"]["ms"],
          lid["ms_per_scan"], rgb["ms_per_scan"], lid["latency_under_solve_us"]["p50"], lid["latency_under_solve_us"]["p99"],
          rgb["latency_under_solve_us"]["p50"], rgb["latency_under_solve_us"]["p99"], lid["latency_under_solve_us"]["over_1ms"],
          rgb["latency_under_solve_us"]["over_1ms"], len(l)))
PY
done
