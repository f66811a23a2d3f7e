#!/usr/bin/env python3
"""What differs between a SLOW and a FAST placement of the materialising pass's three output arrays?
One process: the config-3 batch, VGX_PROBE_SETS sets of (residuals, jac_ref, jac_read) kept alive together, each timed;
then LABELLED launches alternating between the fastest and the slowest set.  Run under `rocprofv3 --pmc ... --kernel-trace
--kernel-include-regex reg_eval_points_kernel` (profiles/placement_pmc.sh): the labelled launches are the LAST dispatches of
that kernel in the process, in the order this script writes to $VGX_PROBE_OUT (labels json), so the counter rows of a pass
can be told apart by set -- slow against fast inside ONE process, on the same box, minutes apart at most."""
import json
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
n_sets = int(os.environ.get("VGX_PROBE_SETS", "6"))
sets = [(torch.empty(R, dtype=torch.float32, device="cuda"), torch.empty((R, 4), dtype=torch.float32, device="cuda"),
         torch.empty((R, 4), dtype=torch.float32, device="cuda")) for _ in range(n_sets)]
torch.cuda.synchronize()
launches = 0


def launch(s):
    global launches
    batch.evaluate_points(poses, s[0].data_ptr(), s[1].data_ptr(), s[2].data_ptr())
    launches += 1


def timed(s, reps=4):
    launch(s)
    ctx.synchronize()
    ctx.timer_start()
    for _ in range(reps):
        launch(s)
    return ctx.timer_stop() / reps


ms = [[timed(s) for s in sets] for _ in range(2)]
best = [min(ms[0][k], ms[1][k]) for k in range(n_sets)]
fast, slow = int(min(range(n_sets), key=lambda k: best[k])), int(max(range(n_sets), key=lambda k: best[k]))
ctx.synchronize()
before = launches
labels = []
for rep in range(int(os.environ.get("VGX_PROBE_LABELLED", "4"))):
    for name, k in (("fast", fast), ("slow", slow)):
        launch(sets[k])
        ctx.synchronize()
        labels.append(name)
# the same two sets timed once more AFTER the labelled launches (did anything drift?)
after = {"fast": timed(sets[fast]), "slow": timed(sets[slow])}
out = {"residuals": int(R), "sets": n_sets, "ms_round0": ms[0], "ms_round1": ms[1], "fast_set": fast, "slow_set": slow,
       "fast_ms": best[fast], "slow_ms": best[slow], "ms_after": after,
       "main_kernel_dispatches_before_labelled": before, "labelled": labels, "main_kernel_dispatches_total": launches,
       "pointers": {name: [hex(t.data_ptr()) for t in sets[k]] for name, k in (("fast", fast), ("slow", slow))}}
path = os.environ.get("VGX_PROBE_OUT")
if path:
    json.dump(out, open(path, "w"))
print(json.dumps(out))
