#!/usr/bin/env python3
"""WHERE a racing scan's latency under a running fused solver evaluation goes: counted scans leave per-workgroup
wall_clock64 stamps (vgx_tsdf_integrator_read_trace).  If the workgroups' starts are spread over the solver kernel's
duration the scan is starved workgroup by workgroup; if they start together and late, the scan's kernel was not
dispatched at all until the solver's kernel had drained."""
import os
import sys
import threading
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from harness.bench_tsdf import sensor_cases, session_scans  # noqa: E402
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
reg_stream = torch.cuda.ExternalStream(ctx.get_stream())


def fused_device_step():
    batch.evaluate_normal(poses, to_host=False)
    reg_stream.synchronize()


for _ in range(3):
    fused_device_step()
dirs, vs, kw, bmin, bdim = sensor_cases()["lidar_64x1024_0.20m_voxgraph_yaml"]
T, clouds = session_scans(dirs, 20)
n_pts = clouds[0].shape[0]
layer = capi.TsdfLayer(ctx, vs, 16)
for k in (0, 19):
    layer.reserve(T[k][4:7], kw["max_ray_length_m"] + kw["default_truncation_distance"] + 2 * vs)
integ = capi.FastTsdfIntegrator(ctx, capi.tsdf_config(**kw), layer)
dev = [torch.from_numpy(c).cuda() for c in clouds]
torch.cuda.synchronize()
for _ in range(5):
    for k in range(20):
        integ.integrate_device(T[k], dev[k].data_ptr(), None, n_pts)
ctx.synchronize()


def scans(label, n=12):
    for k in range(n):
        time.sleep(0.03)
        t0 = time.perf_counter()
        integ.integrate_device(T[k % 20], dev[k % 20].data_ptr(), None, n_pts, count=True)   # (counted: waits for the scan itself)
        lat = (time.perf_counter() - t0) * 1e6
        rows = integ.read_trace(4096)
        starts, ends = rows[:, 0], rows[:, 3]
        print("%s scan %2d: host latency %7.0f us | %d workgroups, first start -> last start %7.1f us, first start -> last end %7.1f us, "
              "median workgroup %5.1f us" % (label, k, lat, len(rows), starts.max() - starts.min(), ends.max() - starts.min(),
                                             float(np.median(ends - starts))))


scans("alone      ")
stop = threading.Event()


def solver():
    while not stop.is_set():
        fused_device_step()


th = threading.Thread(target=solver)
th.start()
time.sleep(0.1)
scans("under fused")
stop.set()
th.join()
