#!/usr/bin/env python3
"""Does a smaller fused-pass tile help the 1/8 shards without costing the single batch?  (VGX_FUSED_TILE_ITERS is
read once per process: run once per value)   gpurun -- 'for t in 0 8 12; do VGX_FUSED_TILE_ITERS=$t python profiles/probes/tile_size_shards.py; done'"""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from voxgraph_amd import capi  # noqa: E402
import torch  # noqa: E402

capi.load()
ctx = capi.Context(0)
a = types.SimpleNamespace(grid=[20, 10], block_dims=[16, 16, 16], block_min=[-8, -8, -4], voxel_size=0.2,
                          truncation=0.6, esdf_max=2.0, pose_sigma=0.3, yaw_sigma=0.05, seed=2)
true_poses, poses, pairs = bench.build_graph(a)
subs, n_points = [], []
for k in range(len(true_poses)):
    sm = capi.Submap.synth_city(ctx, k, 0.2, 16, a.block_min, a.block_dims, 0.6, 2.0, 10.0, true_poses[k], 2)
    n_points.append(sm.extract_voxel_points(1.0, 0.3, True))
    sm.release_raw_layers()
    subs.append(sm)
cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS)
cfs = [capi.RegistrationCostFunction(ctx, subs[i], subs[j], cfg) for i, j in pairs]
full = capi.RegistrationBatch(ctx, cfs, pairs)
w = 36 * np.array([n_points[i] for i, _ in pairs], np.int64) + 45 * full.count_live_each(poses)


def t(bt, reps=10):
    for _ in range(3):
        bt.evaluate_normal(poses, to_host=False)
    ctx.synchronize()
    ctx.timer_start()
    for _ in range(reps):
        bt.evaluate_normal(poses, to_host=False)
    return ctx.timer_stop() / reps


single = t(full)
shard_of = capi.contiguous_shards(w, 8)
ts = []
for r in range(8):
    mine = np.flatnonzero(shard_of == r).astype(np.int32)
    bt = capi.RegistrationBatch(ctx, [cfs[c] for c in mine], pairs[mine], global_index=mine, n_global=len(pairs))
    ts.append(t(bt))
    bt.destroy()
print("VGX_FUSED_TILE_ITERS", os.environ.get("VGX_FUSED_TILE_ITERS", "0 (rule)"), "single ms %.4f" % single,
      "slowest 1/8 shard ms %.4f" % max(ts), "predicted efficiency at N = 8: %.3f" % (single / (8 * max(ts))))
