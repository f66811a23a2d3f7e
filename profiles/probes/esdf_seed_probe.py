#!/usr/bin/env python3
"""One seed of profiles/fuzz_esdf.py in detail: where the device ESDF and the restated queue differ, and by how much."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
F = np.float32
from oracle import pyoracle as orc
from oracle import synth
from voxgraph_amd import capi
capi.load()
ctx = capi.Context(0)
seed = int(os.environ.get("SEED", "5840"))
rng = np.random.default_rng(seed)
vps = int(rng.choice([8, 16])); vs = float(rng.choice([0.05, 0.1, 0.2]))
dims = tuple(int(x) for x in rng.integers(1, 5, 3)); ext = np.array(dims) * vps * vs
c = rng.uniform(0.2, 0.8, 3) * ext
sdf = synth.sphere_ground_sdf(tuple(c), float(rng.uniform(0.2, 0.6) * ext.min()), float(rng.uniform(0.1, 0.4) * ext[2]))
sm = synth.make_submap(sdf, vs, vps, tuple(int(x) for x in rng.integers(-2, 2, 3)), dims, trunc=3 * vs, esdf_max=10 * vs, drop_empty_blocks=bool(rng.integers(0, 2)))
td = sm.tsdf_distance.copy()
if rng.integers(0, 2):
    td = np.clip(td + rng.normal(0, 0.02 * vs, td.shape).astype(F), -3 * vs, 3 * vs).astype(F)
tw = sm.tsdf_weight.copy()
if rng.integers(0, 2):
    tw = np.where(rng.uniform(size=tw.shape) < 0.03, 0, tw).astype(F)
max_d = float(rng.choice([2.0, 6 * vs, 12 * vs]))
kw = dict(max_distance_m=max_d, default_distance_m=float(rng.choice([max_d, 2.0])), min_distance_m=float(rng.choice([0.2, vs, 2 * vs])))
print(dict(seed=seed, vps=vps, vs=vs, dims=dims, **kw))
g = capi.Submap(ctx, 0, vs, vps, sm.block_index, td, tw, None, None)
g.generate_esdf(capi.esdf_config(**kw))
_, _, ed, eo = g.download_layers(vps)
od, oo, _ = orc.esdf_from_tsdf(vs, vps, sm.block_index, td, tw, orc.esdf_config(**kw))
obs = oo.astype(bool)
diff = np.abs(ed - od) * obs
bad = np.argwhere(diff > 2.5e-3)
print("observed equal", np.array_equal(eo, oo), "voxels differing by > 2.5 mm:", len(bad), "of", int(obs.sum()))
for b, v in bad[:12]:
    print(" block", sm.block_index[b], "voxel", v, "tsdf", td[b, v], "w", tw[b, v], "device", ed[b, v], "queue", od[b, v])
vals = np.unique(np.round(np.abs(ed[obs][np.abs(ed[obs] - od[obs]) > 2.5e-3]), 3))[:10], np.unique(np.round(np.abs(od[obs][np.abs(ed[obs] - od[obs]) > 2.5e-3]), 3))[:10]
print("device |d| values at the differing voxels:", vals[0], " queue |d| values:", vals[1])
