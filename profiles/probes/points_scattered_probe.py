#!/usr/bin/env python3
"""Output arrays with their physical chunks mapped in a SHUFFLED order (vgx_bench_alloc_scattered: virtual-memory API, one
physical allocation per chunk) against plain hipMalloc draws: is a deliberately scattered array a reliably fast one for the
materialising kernel (profiles/r05_points_placement.txt: one physical run is the slowest case)?"""
import os
import sys
import time
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


def timed(rp, jop, jep, reps=10):
    for _ in range(2):
        batch.evaluate_points(poses, rp, jop, jep)
    ctx.synchronize()
    ctx.timer_start()
    for _ in range(reps):
        batch.evaluate_points(poses, rp, jop, jep)
    return ctx.timer_stop() / reps


def rate(src, rb, dst, wb):
    return (rb + wb) / capi.stream_ceiling_ms(ctx, src, rb, dst, wb, 3) / 1e6


n16 = (16 * R) & ~15
MB = 1 << 20
chunk = int(os.environ.get("VGX_PROBE_CHUNK_MIB", "32")) * MB
keep = []
for k in range(int(os.environ.get("VGX_PROBE_SETS", "6"))):
    r = capi.alloc_scattered(ctx, 4 * R, chunk, 0); jo = capi.alloc_scattered(ctx, 16 * R, chunk, 0); je = capi.alloc_scattered(ctx, 16 * R, chunk, 0)
    print("set %d  virtual-memory API, %d MiB chunks in order:  kernel %.4f ms" % (k, chunk // MB, timed(r, jo, je)), flush=True)
    t = (torch.empty(R, dtype=torch.float32, device="cuda"), torch.empty((R, 4), dtype=torch.float32, device="cuda"), torch.empty((R, 4), dtype=torch.float32, device="cuda"))
    keep.append(t)
    print("set %d  hipMalloc:                                    kernel %.4f ms" % (k, timed(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr())), flush=True)
    # mixed: the Jacobians from the API, the residuals from hipMalloc, and the other way round
    print("set %d  Jacobians from the API, residuals hipMalloc:  kernel %.4f ms;  the other way round %.4f ms" % (
        k, timed(t[0].data_ptr(), jo, je), timed(r, t[1].data_ptr(), t[2].data_ptr())), flush=True)
