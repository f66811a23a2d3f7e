#!/usr/bin/env python3
"""reg_eval_points_kernel on config 3 (and the full-overlap workload's plain order is not needed here): ms per launch, three
series of 20.  The library comes from VGX_LIB (A/B of two builds on one lease: alternate the two in the calling script)."""
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
r = torch.empty(R, dtype=torch.float32, device="cuda"); jo = torch.empty((R, 4), dtype=torch.float32, device="cuda"); je = torch.empty((R, 4), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
out = []
for s in range(3):
    for _ in range(3):
        batch.evaluate_points(poses, r.data_ptr(), jo.data_ptr(), je.data_ptr())
    ctx.synchronize()
    ctx.timer_start()
    for _ in range(20):
        batch.evaluate_points(poses, r.data_ptr(), jo.data_ptr(), je.data_ptr())
    out.append(ctx.timer_stop() / 20)
chk = float(r[:: 4099].double().sum().item()) + float(jo[:: 4099].double().sum().item()) + float(je[:: 4099].double().sum().item())
print("%s  ms per launch %s  checksum %.9e" % (os.environ.get("VGX_LIB", "tree"), " ".join("%.4f" % x for x in out), chk))
