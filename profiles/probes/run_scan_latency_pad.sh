cd ${GRAFT_REPO_ROOT:-$(pwd)}
for pad in 0 27000 33000 41000; do
  echo "=== VGX_FUSED_LDS_PAD=$pad"
  VGX_FUSED_LDS_PAD=$pad python profiles/probes/scan_latency_probe.py 2>&1 | grep "under fused solver evaluations (device"
done
