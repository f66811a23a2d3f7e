#!/bin/bash
# the fused kernel's gradient from the reference's interp_table_ coefficients (nine FMAs, round 6) against the nested lerps it
# replaced (a library built from the commit before: make SUFFIX=_oldgrad on the stashed tree), alternated twice on one lease
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
ARGS="--steps 10 --warmup 2 --placement-candidates 1 --no-cpu-baseline --no-solve --no-tsdf --no-config5 --no-config2 --no-multi-ctx --no-parity --no-fo-plain"
for rep in 1 2; do
for lib in libvoxgraph_amd.so libvoxgraph_amd_oldgrad.so; do
  VGX_LIB=$REPO/voxgraph_amd/lib/$lib python bench.py $ARGS --detail /tmp/ab_grad.json > /dev/null 2> /tmp/ab_grad.err
  python - "$lib" <<'PY'
import json, sys
d = json.load(open("/tmp/ab_grad.json"))
f, fo, sh = d["fused"], d["roofline_full_overlap"], d.get("shipped_config") or {}
print("%-28s fused c3 %.3f (cost only %.3f)  full overlap %.3f (cost only %.3f)  shipped %.3f ms | cost vs materialised %.2e / %.2e" % (
    sys.argv[1], f["stream_ms_per_step"], f["cost_only_ms"], fo["fused"]["stream_ms_per_step"], fo["fused"]["cost_only_ms"],
    sh.get("stream_ms_per_evaluation") or 0.0, f.get("cost_vs_materialised") or 0.0, fo["fused"].get("cost_vs_materialised") or 0.0))
PY
done
done
