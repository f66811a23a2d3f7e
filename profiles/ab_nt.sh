#!/bin/bash
# interleaved A/B of non-temporal hints in the REG kernels (same box, same process setup):
#   VGX_NT_STORES: output rows of the materialising kernel
#   VGX_NT_LOADS : the 20-byte registration-point stream (both kernels)
for rep in 1 2 3; do
  for cfg in "1 0" "1 1" "0 0"; do
    set -- $cfg
    VGX_NT_STORES=$1 VGX_NT_LOADS=$2 python bench.py --full-line --no-cpu-baseline --no-tsdf --no-solve --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('stores_nt=$1 loads_nt=$2 rep=$rep kernel_ms', round(d['roofline']['kernel_ms'],3), 'frac', round(d['roofline']['frac'],3), 'fused_ms', round(d['fused']['ms_per_step'],3))"
  done
done
