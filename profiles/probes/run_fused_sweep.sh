#!/bin/bash
# the fused pass's experiment switches swept again on the round-6 kernel (lighter arithmetic: has an optimum moved?)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
ARGS="--steps 10 --warmup 2 --placement-candidates 1 --no-cpu-baseline --no-solve --no-tsdf --no-config5 --no-config2 --no-multi-ctx --no-parity --no-fo-plain"
one() {
  env "$@" python bench.py $ARGS --detail /tmp/sweep.json > /dev/null 2> /tmp/sweep.err
  python - "$*" <<'PY'
import json, sys
d = json.load(open("/tmp/sweep.json"))
f, fo, sh = d["fused"], d["roofline_full_overlap"], d.get("shipped_config") or {}
print("%-34s fused c3 %.3f (cost only %.3f)  full overlap %.3f (cost only %.3f)  shipped %.3f ms" % (
    sys.argv[1], f["stream_ms_per_step"], f["cost_only_ms"], fo["fused"]["stream_ms_per_step"], fo["fused"]["cost_only_ms"],
    sh.get("stream_ms_per_evaluation") or 0.0))
PY
}
one VGX_NOP=1
for t in 6 8 12 16 20; do one VGX_FUSED_TILE_ITERS=$t; done
for k in 522 722 822 632; do one VGX_FUSED_KERNEL=$k; done
for w in 48 64 128 192; do one VGX_DRAW_WGS_PER_XCD=$w; done
one VGX_NOP=2
