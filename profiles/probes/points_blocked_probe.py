#!/usr/bin/env python3
"""The materialising pass into ONE array of tile blocks (vgx_reg_batch_evaluate_points_blocked) against three arrays
(vgx_reg_batch_evaluate_points), config 3, several allocations of each in one process, all kept alive: is one write front
as insensitive to where it lies as profiles/r05_points_placement.txt suggests, and how fast?"""
import os
import sys
import types

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
subs = []
for k in range(len(true_poses)):
    sm = capi.Submap.synth_city(ctx, k, 0.2, 16, a.block_min, a.block_dims, 0.6, 2.0, 10.0, true_poses[k], 2)
    sm.extract_voxel_points(1.0, 0.3, True)
    sm.release_raw_layers()
    subs.append(sm)
cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS)
cfs = [capi.RegistrationCostFunction(ctx, subs[i], subs[j], cfg) for i, j in pairs]
batch = capi.RegistrationBatch(ctx, cfs, pairs)
R = batch.num_residuals()
nbytes, rows, first = batch.blocked_layout()
print("residuals %d, blocked array %.3f GB (%.2f %% padding)" % (R, nbytes / 1e9, 100.0 * (nbytes / 36.0 / R - 1.0)))


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    ctx.synchronize()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    return ctx.timer_stop() / reps


keep = []
for k in range(int(os.environ.get("VGX_PROBE_SETS", "6"))):
    blk = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda")
    t = (torch.empty(R, dtype=torch.float32, device="cuda"), torch.empty((R, 4), dtype=torch.float32, device="cuda"), torch.empty((R, 4), dtype=torch.float32, device="cuda"))
    keep.append((blk, t))
    tb = timed(lambda: batch.evaluate_points_blocked(poses, blk.data_ptr()))
    t3 = timed(lambda: batch.evaluate_points(poses, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr()))
    print("allocation %d:  one array of tile blocks %.4f ms    three arrays %.4f ms" % (k, tb, t3), flush=True)
for rnd in range(2):
    print("again:", "  ".join("%.4f / %.4f" % (timed(lambda: batch.evaluate_points_blocked(poses, blk.data_ptr())),
                                                timed(lambda: batch.evaluate_points(poses, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr())))
                              for blk, t in keep), flush=True)
