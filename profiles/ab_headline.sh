#!/bin/bash
# HEAD and the round-2 tree (git archive 98ec687 -> build/r02tree, built with its own Makefile)
# alternately on ONE lease, three times each: is the 4.5 -> 5.4 ms of reg_eval_points_kernel a regression or the box?
#   /usr/local/graft/bin/gpurun --timeout 1500 -- bash profiles/ab_headline.sh
# -> gpurun_out/r05_ab/*.json|*.line ; profiles/ab_headline_summary.py prints the table committed as r05_headline_ab.txt
OUT=gpurun_out/r05_ab
mkdir -p $OUT
python harness/box_state.py > $OUT/box_idle.json 2>&1
COMMON="--steps 20 --warmup 3 --inner 25 --no-cpu-baseline --no-tsdf --no-solve --no-config5 --no-config2 --no-fused --no-full-overlap"
for i in 1 2 3; do
  timeout 300 python bench.py $COMMON --no-parity --detail $OUT/head_$i.json > $OUT/head_$i.line 2> $OUT/head_$i.err
  timeout 300 python build/r02tree/bench.py $COMMON > $OUT/r02_$i.line 2> $OUT/r02_$i.err
done
# (c) candidates on HEAD: the 21 GB of row buffers straight from hipMalloc instead of torch's caching allocator; no
# isosurface extraction before the headline (a different allocation history)
PYTORCH_NO_HIP_MEMORY_CACHING=1 PYTORCH_NO_CUDA_MEMORY_CACHING=1 timeout 300 python bench.py $COMMON --no-parity \
  --detail $OUT/head_nocache.json > $OUT/head_nocache.line 2> $OUT/head_nocache.err
timeout 300 python bench.py $COMMON --no-parity --no-shipped --detail $OUT/head_noshipped.json > $OUT/head_noshipped.line 2> $OUT/head_noshipped.err
# and the driver's own command
timeout 600 python bench.py --detail $OUT/default.json > $OUT/default.line 2> $OUT/default.err
tail -c 600 $OUT/default.line
