#!/bin/bash
# fused-pass tile size, round 5: HEAD~ library (20 Ki tiles, plain butterfly epilogue) against the tree's (10 Ki tiles,
# reduce-scatter epilogue) and the tree's forced back to 20 Ki -- one GPU and the slowest 1/8 shard
#   needs voxgraph_amd/lib/libvoxgraph_amd_head.so (git stash; make -C voxgraph_amd/csrc SUFFIX=_head; git stash pop)
timeout 900 python -m pytest tests/test_reg_gpu.py tests/test_multi_gpu.py tests/test_fuzz_gpu.py tests/test_batch_sampling_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > gpurun_out/pytest_tile.txt 2>&1; grep -E "passed|failed" gpurun_out/pytest_tile.txt
for rep in 1 2; do
[ -f voxgraph_amd/lib/libvoxgraph_amd_head.so ] && VGX_LIB=$PWD/voxgraph_amd/lib/libvoxgraph_amd_head.so timeout 200 python profiles/probes/tile_size_shards.py 2>/dev/null | tail -1 | sed 's/^/round-4 library: /'
timeout 200 python profiles/probes/tile_size_shards.py 2>/dev/null | tail -1 | sed 's/^/this tree:       /'
VGX_FUSED_TILE_ITERS=20 timeout 200 python profiles/probes/tile_size_shards.py 2>/dev/null | tail -1 | sed 's/^/this tree:       /'
done
