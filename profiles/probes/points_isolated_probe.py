#!/usr/bin/env python3
"""reg_eval_points_kernel on config 3: back-to-back launches (what bench.py's timed region is) against launches with the
stream drained in between -- is the 4.4 ms (un-profiled) vs 5.4 ms (under rocprofv3, which serialises dispatches) of
profiles/r05_headline_ab.txt the kernel's own tail overlapping the next launch?"""
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
# Does the kernel's time depend on WHERE the three output arrays lie?  (three separate torch allocations; before them a
# pre-allocation of VGX_PROBE_PRE GiB that stays alive, or is freed again with VGX_PROBE_PRE_FREE=1)


def timed(r_ptr, jo_ptr, je_ptr, reps=20):
    for _ in range(3):
        batch.evaluate_points(poses, r_ptr, jo_ptr, je_ptr)
    ctx.synchronize()
    ctx.timer_start()
    for _ in range(reps):
        batch.evaluate_points(poses, r_ptr, jo_ptr, je_ptr)
    return ctx.timer_stop() / reps


def show(tag, r_ptr, jo_ptr, je_ptr):
    ms = timed(r_ptr, jo_ptr, je_ptr)
    print("%-44s r %#x jo %#x je %#x  (jo - r %+.3f GB, je - jo %+.3f GB)  %.4f ms" % (tag, r_ptr, jo_ptr, je_ptr, (jo_ptr - r_ptr) / 1e9, (je_ptr - jo_ptr) / 1e9, ms))


# (1) three allocations, in the order r, jo, je and in the order je, jo, r
r = torch.empty(R, dtype=torch.float32, device="cuda"); jo = torch.empty((R, 4), dtype=torch.float32, device="cuda"); je = torch.empty((R, 4), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
show("separate, allocated r, jo, je", r.data_ptr(), jo.data_ptr(), je.data_ptr())
del r, jo, je
torch.cuda.empty_cache()
je = torch.empty((R, 4), dtype=torch.float32, device="cuda"); jo = torch.empty((R, 4), dtype=torch.float32, device="cuda"); r = torch.empty(R, dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
show("separate, allocated je, jo, r", r.data_ptr(), jo.data_ptr(), je.data_ptr())
del r, jo, je
torch.cuda.empty_cache()
# (2) one allocation, carved ascending (r, jo, je) and descending (je, jo, r)
big = torch.empty(R * 36 + (8 << 20), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
b0 = big.data_ptr()
al = lambda x: (x + 255) // 256 * 256
show("one allocation, carved r < jo < je", b0, al(b0 + 4 * R), al(al(b0 + 4 * R) + 16 * R))
show("one allocation, carved je < jo < r", al(al(b0 + 16 * R) + 16 * R), al(b0 + 16 * R), b0)
show("one allocation, carved jo < r < je", al(b0 + 16 * R), b0, al(al(b0 + 16 * R) + 4 * R))
del big
torch.cuda.empty_cache()
# (3) three hipMalloc-sized allocations through torch with the caching allocator told to release: again (1), after (2)
r = torch.empty(R, dtype=torch.float32, device="cuda"); jo = torch.empty((R, 4), dtype=torch.float32, device="cuda"); je = torch.empty((R, 4), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
show("separate again, allocated r, jo, je", r.data_ptr(), jo.data_ptr(), je.data_ptr())
