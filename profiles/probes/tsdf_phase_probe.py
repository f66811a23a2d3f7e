#!/usr/bin/env python3
"""Where a racing scan's workgroups spend their time (counted scans leave four wall_clock64 stamps per workgroup: start,
rays queued, walk done, end): median / p90 / max of phase 1, the walk and the folds, and the kernel's span, for a
session-old integrator on the two BASELINE sensor shapes."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from harness.bench_tsdf import sensor_cases, session_scans  # noqa: E402
from voxgraph_amd import capi  # noqa: E402
import torch  # noqa: E402

capi.load()
ctx = capi.Context(0)
for name, (dirs, vs, kw, bmin, bdim) in sensor_cases().items():
    T, clouds = session_scans(dirs, 20)
    n_pts = clouds[0].shape[0]
    layer = capi.TsdfLayer(ctx, vs, 16)
    for k in (0, 19):
        layer.reserve(T[k][4:7], kw["max_ray_length_m"] + kw["default_truncation_distance"] + 2 * vs)
    integ = capi.FastTsdfIntegrator(ctx, capi.tsdf_config(**kw), layer)
    width = int(os.environ.get("WIDTH", "0"))
    if width:
        integ.set_cloud_width(640 if name.startswith("rgbd") else 1024)
    dev = [torch.from_numpy(c).cuda() for c in clouds]
    torch.cuda.synchronize()
    for _ in range(5):
        for k in range(20):
            integ.integrate_device(T[k], dev[k].data_ptr(), None, n_pts)
    ctx.synchronize()
    rows = []
    kern = []
    for k in range(20):
        ctx.synchronize()
        ctx.timer_start()
        integ.integrate_device(T[k], dev[k].data_ptr(), None, n_pts, count=True)
        kern.append(ctx.timer_stop() * 1e3)
        r = integ.read_trace(8192)
        rows.append(r)
    # the uncounted kernel's time for comparison
    unc = []
    for k in range(20):
        ctx.synchronize()
        ctx.timer_start()
        integ.integrate_device(T[k], dev[k].data_ptr(), None, n_pts)
        unc.append(ctx.timer_stop() * 1e3)
    ctx.synchronize()
    ctx.timer_start()
    for _ in range(5):
        for k in range(20):
            integ.integrate_device(T[k], dev[k].data_ptr(), None, n_pts)
    b2b = ctx.timer_stop() * 1e3 / 100
    r = np.concatenate(rows)
    p1, walk, fold, life = r[:, 1] - r[:, 0], r[:, 2] - r[:, 1], r[:, 3] - r[:, 2], r[:, 3] - r[:, 0]
    span = [x[:, 3].max() - x[:, 0].min() for x in rows]
    q = lambda a: "%.1f / %.1f / %.1f" % (np.median(a), np.percentile(a, 90), a.max())
    print(f"{name}: kernel (hip events) counted {np.median(kern):.1f} us, uncounted {np.median(unc):.1f} us, back to back {b2b:.1f} us; span first start -> last end "
          f"{np.median(span):.1f} us; per workgroup (median / p90 / max, us): phase 1 {q(p1)}, walk {q(walk)}, folds {q(fold)}, "
          f"whole {q(life)}; rays per workgroup {q(r[:, 4])}, rounds {q(r[:, 5])}, folds {q(r[:, 6])}")
    integ.destroy()
    layer.destroy()
