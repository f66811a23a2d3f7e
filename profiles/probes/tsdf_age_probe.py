#!/usr/bin/env python3
"""Per-scan time against the integrator's AGE (scans integrated since it was created): voxblox's ApproxHashSet never
forgets between resets -- a mark left by scan N at slot h + N reads as "present" for the voxel with hash h - k in scan N + k --
so rays of later scans are cut short by marks of earlier ones, and a session-old integrator (the reference keeps ONE for the
whole mapping session: pointcloud_integrator.cpp:66-75) does less work per scan than a fresh one.  The bench sessions repeated
12 times through one integrator; racing mode = kernel by itself (stream drained), reproducible / merged = wall clock."""
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from harness.bench_tsdf import sensor_cases, session_scans  # noqa: E402


def main(scans=20, passes=12):
    import torch
    from voxgraph_amd import capi
    capi.load()
    ctx = capi.Context(0)
    out = {}
    for name, (dirs, vs, kw, _, _) in sensor_cases().items():
        poses, clouds = session_scans(dirs, scans)
        n_pts = clouds[0].shape[0]
        reach = kw["max_ray_length_m"] + kw["default_truncation_distance"] + 2 * vs
        dev = [torch.from_numpy(c_).cuda() for c_ in clouds]
        torch.cuda.synchronize()

        def new_layer():
            lay_ = capi.TsdfLayer(ctx, vs, 16)
            for k in (0, scans - 1):
                lay_.reserve(poses[k][4:7], reach)
            return lay_
        res = {}
        gc.collect()
        gc.disable()
        for mode in ("racing", "reproducible", "merged"):
            lay = new_layer()
            cfg = capi.tsdf_config(deterministic=1, **kw) if mode == "reproducible" else capi.tsdf_config(**kw)
            integ = capi.FastTsdfIntegrator(ctx, cfg, lay)
            call = integ.integrate_merged_device if mode == "merged" else integ.integrate_device
            rows = []
            for p in range(passes if mode != "merged" else 3):
                if mode == "racing":
                    per = []
                    for k in range(scans):
                        ctx.synchronize()
                        ctx.timer_start()
                        call(poses[k], dev[k].data_ptr(), None, n_pts)
                        per.append(ctx.timer_stop() * 1e3)
                    iso = float(np.median(per[1:]))
                else:
                    iso = None
                # back to back, a fresh layer each pass (what bench.py's ms_per_scan is)
                lay2 = new_layer()
                integ.setLayer(lay2)
                call(poses[0], dev[0].data_ptr(), None, n_pts)
                ctx.synchronize()
                t0 = time.perf_counter()
                for k in range(1, scans):
                    call(poses[k], dev[k].data_ptr(), None, n_pts)
                ctx.synchronize()
                b2b = (time.perf_counter() - t0) * 1e6 / (scans - 1)
                u = call(poses[1], dev[1].data_ptr(), None, n_pts, count=True)
                lay.destroy()
                lay = lay2
                rows.append({"age_at_start": p * (2 * scans + 1) if mode == "racing" else p * (scans + 1), "kernel_us_median": iso and round(iso, 1),
                             "back_to_back_us": round(b2b, 1), "updates_scan1": int(u)})
            res[mode] = rows
            integ.destroy()
            lay.destroy()
        gc.enable()
        out[name] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
