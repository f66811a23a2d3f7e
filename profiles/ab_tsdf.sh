#!/bin/bash
# A/B of the software-pipelined voxel updates in tsdf_integrate_kernel (VGX_TSDF_PIPELINED=0/1),
# interleaved on one box: the `tsdf` object of the bench line (RGB-D and LiDAR scans).
for rep in 1 2; do
for p in 0 1; do
VGX_TSDF_PIPELINED=$p python bench.py --full-line --no-cpu-baseline --no-solve --no-fused --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())['tsdf']
print('rep=$rep pipelined=$p', {k: (round(v['ms_per_scan'],4), round(v['Mpoints_per_s'])) for k,v in d.items()})"
done
done
