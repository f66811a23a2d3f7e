#!/bin/bash
# profiles/r06_scan_latency.txt: per-scan latency under a running solve, stream priorities on (shipped) and off
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for p in 1 0 1 0; do
  echo "=== VGX_STREAM_PRIORITY=$p"
  VGX_STREAM_PRIORITY=$p python profiles/probes/scan_latency_probe.py 2>&1 | grep -v "^$" | tail -10
done
