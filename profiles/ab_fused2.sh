#!/bin/bash
# Interleaved A/B of the fused-kernel variants (VGX_FUSED_KERNEL, see vgx_reg.hip) and of the
# -fslp-vectorize build of the library (make SUFFIX=_slp EXTRA=-fslp-vectorize), on one box:
#   gpurun -- 'bash profiles/ab_fused2.sh'
# Prints one line per (library, variant): fused ms per solver evaluation on config 3 and on the
# full-overlap workload, its agreement with the materialised sums, and the materialising kernel's
# ms on both workloads.  Two rounds, so box drift shows up as a difference between the rounds.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p $OUT
ARGS="--steps 5 --warmup 1 --inner 2 --no-cpu-baseline --no-solve --no-tsdf --no-shipped --no-config5 --no-config2 --no-multi-ctx --no-parity"
pick='import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1])
f,fo=d["fused"],d["roofline_full_overlap"]
print("fused %.3f ms (stream %.3f, rel.err %.1e) | fused full-overlap %.3f ms (rel.err %.1e) | points %.3f ms | points full-overlap %.3f ms" % (
 f["ms_per_step"],f["stream_ms_per_step"],f["cost_vs_materialised"],fo["fused"]["ms_per_step"],fo["fused"]["cost_vs_materialised"],
 d["roofline"]["kernel_ms"],fo["kernel_ms"]))'
for round in ${ROUNDS:-1 2}; do
  for lib in libvoxgraph_amd.so libvoxgraph_amd_slp.so; do
    [ -f $REPO/voxgraph_amd/lib/$lib ] || continue
    for v in ${VARIANTS:-0 421 422 522 622 612 812}; do
      printf "round %s %-26s VGX_FUSED_KERNEL=%-4s " $round $lib $v
      VGX_LIB=$REPO/voxgraph_amd/lib/$lib VGX_FUSED_KERNEL=$v python $REPO/bench.py --full-line $ARGS 2>$OUT/ab_fused2.err | python -c "$pick" || tail -3 $OUT/ab_fused2.err
    done
  done
done
