#!/usr/bin/env python3
"""Output arrays from hipExtMallocWithFlags(hipDeviceMallocContiguous) -- physically contiguous -- against plain hipMalloc, in
one process, alternating: does the materialising kernel's fast / slow lottery (profiles/r05_points_placement.txt) go away
when the pages of an array are one physical run?"""
import ctypes
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
hip = None
with open("/proc/self/maps") as f:
    for name in sorted({l.split()[-1] for l in f if "libamdhip64" in l}):
        hip = ctypes.CDLL(name)
        break
hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]


def alloc(nbytes, contiguous):
    p = ctypes.c_void_p()
    rc = hip.hipExtMallocWithFlags(ctypes.byref(p), nbytes, 0x4) if contiguous else hip.hipMalloc(ctypes.byref(p), nbytes)
    if rc != 0 or not p.value:
        raise RuntimeError("allocation of %d bytes failed (contiguous=%s): hip error %d" % (nbytes, contiguous, rc))
    return p.value


def timed(rp, jop, jep, reps=10):
    for _ in range(2):
        batch.evaluate_points(poses, rp, jop, jep)
    ctx.synchronize()
    ctx.timer_start()
    for _ in range(reps):
        batch.evaluate_points(poses, rp, jop, jep)
    return ctx.timer_stop() / reps


SL = 1 << 29
r0, jo0, je0 = alloc(4 * R + SL, True), alloc(16 * R + SL, True), alloc(16 * R + SL, True)
print("contiguous: r %#x jo %#x je %#x" % (r0, jo0, je0))
d_r, d_jo, d_je = alloc(4 * R + SL, False), alloc(16 * R + SL, False), alloc(16 * R + SL, False)
n16 = (16 * R) & ~15
h16 = (n16 // 2) & ~15


def rate(src, rb, dst, wb):
    return (rb + wb) / capi.stream_ceiling_ms(ctx, src, rb, dst, wb, 3) / 1e6


for tag, (a_, b_, c_) in (("contiguous", (r0, jo0, je0)), ("default", (d_r, d_jo, d_je))):
    print("%-10s fill jo %.0f  fill je %.0f  read jo %.0f  read je %.0f  copy jo->je %.0f  copy first->second half of je %.0f GB/s   kernel %.4f ms" % (
        tag, rate(None, 0, b_, n16), rate(None, 0, c_, n16), rate(b_, n16, b_, 0), rate(c_, n16, c_, 0), rate(b_, n16, c_, n16),
        rate(c_, h16, c_ + h16, h16), timed(a_, b_, c_)), flush=True)
# every tile culled (poses 10 km apart: no evaluation finds a reading block): the kernel's write pattern by itself
import numpy as np  # noqa: E402
far = np.array(poses, dtype=np.float64).copy()
far[:, 0] += 1e4 * np.arange(len(far))
near = poses


def timed_poses(ps, rp, jop, jep, reps=10):
    for _ in range(2):
        batch.evaluate_points(ps, rp, jop, jep)
    ctx.synchronize()
    ctx.timer_start()
    for _ in range(reps):
        batch.evaluate_points(ps, rp, jop, jep)
    return ctx.timer_stop() / reps


for tag, (a_, b_, c_) in (("contiguous", (r0, jo0, je0)), ("default", (d_r, d_jo, d_je))):
    t_far = timed_poses(far, a_, b_, c_)
    print("%-10s every tile culled: %.4f ms = %.0f GB/s written;  residuals only %.4f ms;  residuals + jac_ref %.4f ms" % (
        tag, t_far, 36.0 * R / t_far / 1e6, timed_poses(far, a_, None, None), timed_poses(far, a_, b_, None)), flush=True)
# mixed: which of the three arrays has to be contiguous for the kernel to be slow?
for tag, (a_, b_, c_) in (("r contiguous", (r0, d_jo, d_je)), ("jac_ref contiguous", (d_r, jo0, d_je)), ("jac_read contiguous", (d_r, d_jo, je0)),
                         ("both Jacobians contiguous", (d_r, jo0, je0))):
    print("%-28s kernel %.4f ms" % (tag, timed(a_, b_, c_)), flush=True)
if os.environ.get("VGX_PROBE_SHIFTS"):
    MB = 1 << 20
    for sh in (0, 64 << 10, 256 << 10, 1 * MB, 2 * MB, 3 * MB, 4 * MB, 6 * MB, 8 * MB, 12 * MB, 16 * MB, 24 * MB, 32 * MB, 48 * MB, 64 * MB, 96 * MB, 128 * MB, 192 * MB, 256 * MB, 384 * MB):
        print("shift %9d B (%6.2f MiB):  jac_read shifted %.4f   jac_ref shifted %.4f   residuals shifted %.4f   both Jacobians shifted %.4f" % (
            sh, sh / MB, timed(r0, jo0, je0 + sh), timed(r0, jo0 + sh, je0), timed(r0 + sh, jo0, je0), timed(r0, jo0 + sh, je0 + sh)), flush=True)
