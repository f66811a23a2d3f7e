#!/bin/bash
# profiles/r06_locate_upper_bound.txt: what could a cheaper point location gain the fused pass?  The shipped library against
# a build whose locate_axis is ONE fma + floor + shifts (no guard, offsets not the reference's bits: an upper bound on
# what a guarded two-speed locate_axis could save).  make SUFFIX=_fastloc EXTRA=-DVGX_LOCATE_UPPER_BOUND
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
ARGS="--steps 10 --warmup 2 --placement-candidates 1 --no-cpu-baseline --no-solve --no-tsdf --no-config5 --no-config2 --no-multi-ctx --no-parity --no-shipped --no-fo-plain"
for rep in 1 2; do
for lib in libvoxgraph_amd.so libvoxgraph_amd_fastloc.so; do
  VGX_LIB=$REPO/voxgraph_amd/lib/$lib python bench.py $ARGS --detail /tmp/ab_locate.json > /dev/null 2> /tmp/ab_locate.err
  python - "$lib" <<'PY'
import json, sys
d = json.load(open("/tmp/ab_locate.json"))
f, fo = d["fused"], d["roofline_full_overlap"]
print("%-28s points c3 %.3f ms  full overlap %.3f ms | fused c3 %.3f (cost only %.3f)  full overlap %.3f (cost only %.3f) ms" % (
    sys.argv[1], d["roofline"]["kernel_ms"], fo["kernel_ms"], f["stream_ms_per_step"], f["cost_only_ms"],
    fo["fused"]["stream_ms_per_step"], fo["fused"]["cost_only_ms"]))
PY
done
done
