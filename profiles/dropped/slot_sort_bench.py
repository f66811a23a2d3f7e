#!/usr/bin/env python3
"""The sort-based TSDF paths' (key, position) sort by itself (vgx_bench_slot_sort, HIP events, 50 sorts) next to
torch.sort(stable=True) on the same keys, at the sizes the bench's scans produce:
    gpurun -- 'python profiles/slot_sort_bench.py'"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from voxgraph_amd import capi
    capi.load()
    ctx = capi.Context(0)
    rng = np.random.default_rng(0)
    for n, bits, what in ((65_536, 21, "LiDAR points by start slot"), (138_000, 20, "merged LiDAR records"),
                          (237_568, 20, "LiDAR accesses"), (307_200, 21, "depth-image points by start slot"),
                          (2_272_768, 20, "depth-image accesses"), (3_323_136, 20, "merged depth-image records")):
        keys = torch.from_numpy(rng.integers(0, 1 << bits, n, dtype=np.uint32).view(np.int32)).cuda()
        out = torch.empty_like(keys)
        idx = torch.empty_like(keys)
        ms = capi.slot_sort(ctx, keys.data_ptr(), n, bits, out.data_ptr(), idx.data_ptr(), repeats=51)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.sort(keys, stable=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(50):
            torch.sort(keys, stable=True)
        e1.record()
        torch.cuda.synchronize()
        print(f"{what:34s} n={n:8d} bits={bits}: {ms * 1e3:7.1f} us per sort, {n * 8 * 2 * ((bits + 7) // 8) / ms / 1e6:7.1f} GB/s "
              f"moved;  torch.sort(stable) {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us")
    ctx.close()


if __name__ == "__main__":
    main()
