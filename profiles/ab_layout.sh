#!/bin/bash
# A/B of the brick layouts (vgx_internal.h) as the DEFAULT of every context: apron (default build), quad
# (make SUFFIX=_quad EXTRA=-DVGX_BRICK_LAYOUT_DEFAULT=1), sub-tiles (make SUFFIX=_sub EXTRA=-DVGX_BRICK_LAYOUT_DEFAULT=2).
# (At run time a context picks apron or quad with vgx_ctx_set_brick_layout; bench.py's
# shipped_config.quad_bricks measures that.)
#   gpurun -- 'bash profiles/ab_layout.sh'
# Per library: the REG parity tests, then fused / materialising ms on config 3 and full overlap and the
# shipped (sampled) configuration's ms per evaluation.  Two rounds (box drift).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
ARGS="--steps 5 --warmup 1 --inner 2 --no-cpu-baseline --no-solve --no-tsdf --no-config5 --no-config2 --no-multi-ctx --no-parity"
pick='import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1])
f,fo,sh=d["fused"],d["roofline_full_overlap"],d["shipped_config"]
print("fused %.3f ms (rel.err %.1e) | fused FO %.3f ms (rel.err %.1e) | points %.3f ms | points FO %.3f (plain %.3f) ms | shipped %.3f ms (stream %.3f)" % (
 f["stream_ms_per_step"],f["cost_vs_materialised"],fo["fused"]["stream_ms_per_step"],fo["fused"]["cost_vs_materialised"],
 d["roofline"]["kernel_ms"],fo["kernel_ms"],fo["plain_order"]["kernel_ms"],sh["ms_per_evaluation"],sh["stream_ms_per_evaluation"]))'
for lib in ${LIBS:-libvoxgraph_amd.so libvoxgraph_amd_quad.so libvoxgraph_amd_sub.so}; do
  [ -f $REPO/voxgraph_amd/lib/$lib ] || continue
  printf "%-26s parity: " $lib
  VGX_LIB=$REPO/voxgraph_amd/lib/$lib python -m pytest $REPO/tests/test_reg_gpu.py $REPO/tests/test_fullsize_gpu.py -m gpu -q -x 2>&1 | tail -1
done
for round in 1 2; do
  for lib in ${LIBS:-libvoxgraph_amd.so libvoxgraph_amd_quad.so libvoxgraph_amd_sub.so}; do
    [ -f $REPO/voxgraph_amd/lib/$lib ] || continue
    printf "round %s %-26s " $round $lib
    VGX_LIB=$REPO/voxgraph_amd/lib/$lib python $REPO/bench.py --full-line $ARGS 2>$OUT/ab_layout.err | python -c "$pick" || tail -3 $OUT/ab_layout.err
  done
done
