#!/bin/bash
# profiles/r06_scan_latency.txt, part 2: WHY a scan waits for the whole fused kernel and not for the materialising one.
#   VGX_FUSED_KERNEL=622 / 522 / 422 : the fused kernel at 6 / 5 / 4 wavefronts per SIMD (is it VGPR space?)
#   VGX_TSDF_KERNEL=v1               : the one-thread-per-point TSDF kernel of rounds 1-4, which uses NO LDS (is it LDS space?)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for v in ${VARIANTS:-622 522 422}; do
  echo "=== VGX_FUSED_KERNEL=$v"
  VGX_FUSED_KERNEL=$v python profiles/probes/scan_latency_probe.py 2>&1 | grep "under"
done
echo "=== VGX_TSDF_KERNEL=v1"
VGX_TSDF_KERNEL=v1 python profiles/probes/scan_latency_probe.py 2>&1 | grep "under"
