import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
F = np.float32
import torch
from oracle import pyoracle as orc
from voxgraph_amd import capi
capi.load()
ctx = capi.Context(0)
seed = 8
vs, vps = 0.2, 16
bmin, bdim = (-4, -4, -2), (8, 8, 8)
rng = np.random.default_rng(seed)
n_sub = 4
true = np.c_[rng.uniform(-14, 14, (n_sub, 2)), rng.uniform(-0.5, 0.5, n_sub), rng.uniform(-0.4, 0.4, n_sub)]
subs, layers, pts = [], [], []
for k in range(n_sub):
    sm = capi.Submap.synth_city(ctx, k, vs, vps, bmin, bdim, 0.6, 2.0, 10.0, true[k], seed % 5)
    n = sm.extract_voxel_points(1.0, 0.3, True)
    td, tw, ed, eo = sm.download_layers(vps)
    layers.append(orc.Layer(vs, vps, sm.block_index(), ed, eo))
    pts.append(sm.download_points(capi.POINTS_VOXELS) if n else None)
    subs.append(sm)
pairs = [(a, b) for a in range(n_sub) for b in range(n_sub) if a != b and pts[a] is not None]
nc = float(rng.choice([0.0, 0.0, 0.25]))
cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS, use_esdf_distance=1, no_correspondence_cost=nc)
cfs = [capi.RegistrationCostFunction(ctx, subs[a], subs[b], cfg) for a, b in pairs]
batch = capi.RegistrationBatch(ctx, cfs, pairs)
poses = true + rng.normal(0, 1, (n_sub, 4)) * [0.3, 0.3, 0.05, 0.05]
_, normal = batch.evaluate_normal(poses)
out = []
for c, (a, b) in enumerate(pairs):
    xyz, dist, w = pts[a]
    ok, r0, jo0, je0 = orc.reg_evaluate(layers[b], xyz, dist, w, poses[a], poses[b], no_correspondence_cost=nc)
    cost = float(r0 @ r0)
    r32 = r0.astype(F).astype(np.float64)
    nz = r0[r0 != 0]
    out.append((abs(normal[c][0] - cost) / max(cost, 1e-300), (a, b), cost, len(nz), float(np.abs(nz).max()) if len(nz) else 0, float(np.median(np.abs(nz))) if len(nz) else 0,
                abs(float(r32 @ r32) - cost) / max(cost, 1e-300)))
for o in sorted(out, reverse=True)[:5]:
    print("fused cost rel err %.3e  pair %s cost %.6g  nonzero rows %d  max|r| %.3g median|r| %.3g   (f32-rounded rows' own sum rel err %.1e)" % o)
print("variant", os.environ.get("VGX_FUSED_KERNEL", "default"))
