#!/bin/bash
# A/B of two builds of the library on one lease (HEAD's in build/headtree, the working tree's in place):
#   /usr/local/graft/bin/gpurun --timeout 900 -- bash profiles/probes/run_points_ab.sh
for k in 1 2 3; do
  VGX_LIB=$PWD/build/headtree/voxgraph_amd/lib/libvoxgraph_amd.so timeout 200 python profiles/probes/points_ab_probe.py 2>/dev/null | tail -1
  timeout 200 python profiles/probes/points_ab_probe.py 2>/dev/null | tail -1
done
