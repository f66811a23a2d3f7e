#!/usr/bin/env python3
"""Drop-in vgx_reg_evaluate called the way Ceres calls it: T threads, each evaluating its own
cost functions (pose_graph.cpp:96 sets num_threads = 4).  Constraint size = a voxgraph-sized
submap pair (config 1: ~6e4 registration points).  Prints evaluations/s for T = 1, 2, 4, 8."""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import synth          # noqa: E402  (scene generator only)
from voxgraph_amd import capi     # noqa: E402

capi.load()
ctx = capi.Context(0)
ref, read = synth.config1_pair(seed=0, asymmetric=True)
mk = lambda sm, i: capi.Submap(ctx, i, sm.voxel_size, sm.vps, sm.block_index, sm.tsdf_distance,
                               sm.tsdf_weight, sm.esdf_distance, sm.esdf_observed)
a, b = mk(ref, 0), mk(read, 1)
a.extract_voxel_points(1.0, 0.3, True)
cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS)
NCF = 8
cfs = [capi.RegistrationCostFunction(ctx, a, b, cfg) for _ in range(NCF)]
N = cfs[0].num_residuals()
bufs = [(np.zeros(N), np.zeros((N, 4)), np.zeros((N, 4))) for _ in range(NCF)]
pa, pb = np.array([0.05, 0.0, 0.02, 0.01]), np.array([0.0, 0.03, 0.0, -0.02])


def work(i, reps):
    r, j0, j1 = bufs[i]
    for _ in range(reps):
        cfs[i].Evaluate([pa, pb], r, [j0, j1])
    return reps


out = {"residuals_per_evaluate": N, "threads": {}}
for T in (1, 2, 4, 8):
    with ThreadPoolExecutor(T) as ex:
        list(ex.map(lambda i: work(i, 5), range(T)))
        t0 = time.perf_counter()
        done = sum(ex.map(lambda i: work(i, 200), range(T)))
        dt = time.perf_counter() - t0
    out["threads"][T] = {"evaluations_per_s": done / dt, "us_per_evaluation": dt / done * 1e6,
                         "Mresiduals_per_s": done * N / dt / 1e6}
ref_r = bufs[0][0].copy()
assert all(np.array_equal(bufs[i][0], ref_r) for i in range(NCF))

# the shipped configuration: sampling_ratio 0.05 (voxgraph_mapper.yaml:34) -- tiny evaluations whose
# cost is the call itself (host-side draws, one small upload, launch, small copies back)
scfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS, sampling_ratio=0.05)
scfs = [capi.RegistrationCostFunction(ctx, a, b, scfg) for _ in range(NCF)]
NS = scfs[0].num_residuals()
sbufs = [(np.zeros(NS), np.zeros((NS, 4)), np.zeros((NS, 4))) for _ in range(NCF)]


def swork(i, reps):
    r, j0, j1 = sbufs[i]
    for _ in range(reps):
        scfs[i].Evaluate([pa, pb], r, [j0, j1])
    return reps


out["sampled_0.05"] = {"residuals_per_evaluate": NS, "threads": {}}
for T in (1, 4):
    with ThreadPoolExecutor(T) as ex:
        list(ex.map(lambda i: swork(i, 5), range(T)))
        t0 = time.perf_counter()
        done = sum(ex.map(lambda i: swork(i, 300), range(T)))
        dt = time.perf_counter() - t0
    out["sampled_0.05"]["threads"][T] = {"evaluations_per_s": done / dt, "us_per_evaluation": dt / done * 1e6,
                                          "Mresiduals_per_s": done * NS / dt / 1e6}
print(json.dumps(out, indent=1))
