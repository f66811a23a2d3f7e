#!/bin/bash
# A/B of the fused kernel's points-per-thread (VGX_REDUCE_PPT) on the bench workload.
for p in 4 2 1; do
  echo "PPT=$p"
  VGX_REDUCE_PPT=$p python bench.py --no-cpu-baseline --no-tsdf --no-solve --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(' value', round(d['value']), 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'fused ms/step', round(d['fused']['ms_per_step'],3), 'fused GB/s', round(d['fused']['algorithmic_GBs']))"
done
