#!/usr/bin/env python3
"""Does the headline kernel's time go with WHERE its three output arrays lie physically?  One process, one batch, eight
sets of (residuals, jac_ref, jac_read) allocated one after the other and all kept alive (so every set has its own physical
pages), the kernel timed on each set three times round-robin.  VGX_LIB selects the build."""
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
sets = []
for k in range(int(os.environ.get("VGX_PROBE_SETS", "8"))):
    SLACK = (1 << 30) // 16 + 4096      # rows of slack: the arrays can be shifted by up to 1 GiB inside their allocations
    r = torch.empty(R + 4 * SLACK, dtype=torch.float32, device="cuda"); jo = torch.empty((R + SLACK, 4), dtype=torch.float32, device="cuda"); je = torch.empty((R + SLACK, 4), dtype=torch.float32, device="cuda")
    sets.append((r, jo, je))
torch.cuda.synchronize()


def timed(s, reps=10):
    r, jo, je = s
    for _ in range(2):
        batch.evaluate_points(poses, r.data_ptr(), jo.data_ptr(), je.data_ptr())
    ctx.synchronize()
    ctx.timer_start()
    for _ in range(reps):
        batch.evaluate_points(poses, r.data_ptr(), jo.data_ptr(), je.data_ptr())
    return ctx.timer_stop() / reps


rows = [[] for _ in sets]
for rnd in range(3):
    for k, s in enumerate(sets):
        rows[k].append(timed(s))
def fill(tensor, nbytes):
    return nbytes / capi.stream_ceiling_ms(ctx, 0, 0, tensor.data_ptr(), nbytes, 3) / 1e6


def read(tensor, nbytes):
    return nbytes / capi.stream_ceiling_ms(ctx, tensor.data_ptr(), nbytes, tensor.data_ptr(), 0, 3) / 1e6


for k, (s, t) in enumerate(zip(sets, rows)):
    f = [fill(s[0], 4 * R), fill(s[1], 16 * R), fill(s[2], 16 * R)]
    g = [read(s[1], 16 * R), read(s[2], 16 * R)]
    cp = 32.0 * R / capi.stream_ceiling_ms(ctx, s[1].data_ptr(), 16 * R, s[2].data_ptr(), 16 * R, 3) / 1e6
    half = (8 * R) & ~15
    self_copy = [2.0 * half / capi.stream_ceiling_ms(ctx, s[w].data_ptr(), half, s[w].data_ptr() + half, half, 3) / 1e6 for w in (1, 2)]
    print("       first half -> second half of jo / of je (two streams inside ONE array): %.0f %.0f GB/s" % tuple(self_copy))
    print("set %d  r %#x jo %#x je %#x   ms %s   fill r/jo/je %.0f %.0f %.0f  read jo/je %.0f %.0f  copy jo->je %.0f GB/s" % (
        k, s[0].data_ptr(), s[1].data_ptr(), s[2].data_ptr(), " ".join("%.4f" % x for x in t), f[0], f[1], f[2], g[0], g[1], cp))
# mixed sets: r/jo of one with je of another
for (i, j) in ((0, 1), (1, 0), (2, 5), (5, 2)):
    if max(i, j) >= len(sets):
        continue
    t = timed((sets[i][0], sets[i][1], sets[j][2]))
    print("r, jo of set %d with je of set %d   ms %.4f" % (i, j, t))

# alignment with the placement held fixed: the fastest set's arrays, the Jacobian arrays shifted by a few bytes inside their
# own allocations (the same physical pages)
best = min(range(len(sets)), key=lambda k: min(rows[k]))
worst = max(range(len(sets)), key=lambda k: min(rows[k]))


def timed_ptrs(rp, jop, jep, reps=10):
    for _ in range(2):
        batch.evaluate_points(poses, rp, jop, jep)
    ctx.synchronize()
    ctx.timer_start()
    for _ in range(reps):
        batch.evaluate_points(poses, rp, jop, jep)
    return ctx.timer_stop() / reps


for tag, k in (("fastest set", best), ("slowest set", worst)):
    r, jo, je = sets[k]
    for sh in (0, 256, 2048, 4096, 8192, 16384 + 2048):
        print("%s (%d), jac_ref and jac_read + %5d B   ms %.4f" % (tag, k, sh, timed_ptrs(r.data_ptr(), jo.data_ptr() + sh, je.data_ptr() + sh)))
    print("%s (%d), residuals + 1024 B                 ms %.4f" % (tag, k, timed_ptrs(r.data_ptr() + 1024, jo.data_ptr(), je.data_ptr())))

# larger shifts of ONE array inside its allocation: is the relation between the arrays periodic in their distance?
for tag, k in (("slowest set", worst), ("fastest set", best)):
    r, jo, je = sets[k]
    for sh in (0, 64 << 10, 1 << 20, 2 << 20, 16 << 20, 128 << 20, 512 << 20, 1 << 30):
        a = timed_ptrs(r.data_ptr(), jo.data_ptr(), je.data_ptr() + sh)
        b = timed_ptrs(r.data_ptr(), jo.data_ptr() + sh, je.data_ptr())
        c = timed_ptrs(r.data_ptr() + sh, jo.data_ptr(), je.data_ptr())
        print("%s (%d), shift %10d B:  jac_read shifted %.4f   jac_ref shifted %.4f   residuals shifted %.4f" % (tag, k, sh, a, b, c))
