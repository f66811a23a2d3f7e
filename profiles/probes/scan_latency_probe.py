#!/usr/bin/env python3
"""Per-scan latency of the racing TSDF mode while a solve runs on the same context: LiDAR 64 x 1024 at
10 Hz and 640 x 480 depth images at 30 Hz on thread A, thread B looping (a) fused solver evaluations or (b) materialising
passes of the config-3 graph.  Run once with VGX_STREAM_PRIORITY=1 (shipped: TSDF stream high, registration stream low) and
once with 0 (both default): profiles/probes/run_scan_latency.sh."""
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from harness.bench_latency import latency_block  # noqa: E402
from harness.bench_tsdf import sensor_cases, session_scans, AGE_PASSES  # noqa: E402
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
r = torch.empty(R, dtype=torch.float32, device="cuda")
jo = torch.empty((R, 4), dtype=torch.float32, device="cuda")
je = torch.empty((R, 4), dtype=torch.float32, device="cuda")
reg_stream = torch.cuda.ExternalStream(ctx.get_stream())


def fused_step():
    batch.evaluate_normal(poses, to_host=True)


def points_step():
    batch.evaluate_points(poses, r.data_ptr(), jo.data_ptr(), je.data_ptr())
    reg_stream.synchronize()


def fused_device_step():
    batch.evaluate_normal(poses, to_host=False)
    reg_stream.synchronize()


for _ in range(3):
    fused_step()
    points_step()
print(json.dumps({"stream_priorities": ctx.stream_priorities(), "VGX_STREAM_PRIORITY": os.environ.get("VGX_STREAM_PRIORITY")}))
for name, (dirs, vs, kw, bmin, bdim) in sensor_cases().items():
    sensor = "rgbd" if name.startswith("rgbd") else "lidar"
    T, clouds = session_scans(dirs, 20)
    n_pts = clouds[0].shape[0]
    layer = capi.TsdfLayer(ctx, vs, 16)
    for k in (0, 19):
        layer.reserve(T[k][4:7], kw["max_ray_length_m"] + kw["default_truncation_distance"] + 2 * vs)
    integ = capi.FastTsdfIntegrator(ctx, capi.tsdf_config(**kw), layer)
    dev = [torch.from_numpy(c).cuda() for c in clouds]
    torch.cuda.synchronize()
    for _ in range(1 + AGE_PASSES):
        for k in range(20):
            integ.integrate_device(T[k], dev[k].data_ptr(), None, n_pts)
    ctx.synchronize()
    hz, n = (30.0, 60) if sensor == "rgbd" else (10.0, 30)
    if os.environ.get("VGX_LAT_TAIL"):
        # the tail by itself: many scans at a faster cadence under the host-blocking fused evaluations; every scan above 1 ms
        # with its index (is it the first ones?  periodic?  how often?)
        from harness.bench_latency import scan_latency
        n_tail = int(os.environ["VGX_LAT_TAIL"])
        for rep in range(2):
            lat, evals = scan_latency(capi, ctx, torch, integ, T, dev, n_pts, 4 * hz, n_tail, fused_step)
            a_ = np.asarray(lat)
            print(sensor, "tail", json.dumps({"scans": n_tail, "cadence_Hz": 4 * hz, "p50": float(np.percentile(a_, 50)), "p99": float(np.percentile(a_, 99)),
                                              "max": float(a_.max()), "over_1ms": [(int(i), round(float(a_[i]))) for i in np.nonzero(a_ > 1000)[0]],
                                              "solver_evaluations": evals}))
        integ.destroy()
        layer.destroy()
        continue
    for load, step in (("fused solver evaluations (blocks to the host)", fused_step),
                       ("fused solver evaluations (device only)", fused_device_step), ("materialising passes", points_step)):
        out = latency_block(capi, ctx, torch, integ, T, dev, n_pts, hz, n, step, None)
        out.pop("what")
        print(sensor, "under", load, json.dumps(out))
    integ.destroy()
    layer.destroy()
