#!/bin/bash
# interleaved A/B of the XCD-aware tile order (swizzle_tile) in the REG kernels, same box
for rep in 1 2 3; do
  for sw in 1 0; do
    VGX_XCD_SWIZZLE=$sw python bench.py --full-line --no-cpu-baseline --no-tsdf --no-solve --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('swizzle=$sw rep=$rep kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],3), 'fused_ms', round(d['fused']['ms_per_step'],3))"
  done
done
