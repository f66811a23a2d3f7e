#!/bin/bash
# repeated fused-pass timing (same box) used for A/Bs of reg_eval_reduce_kernel variants
for rep in 1 2 3; do
python bench.py --full-line --no-cpu-baseline --no-tsdf --no-solve --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('rep=$rep fused_ms', round(d['fused']['ms_per_step'],3), 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'cost_vs_mat', d['fused']['cost_vs_materialised'])"
done
