#!/usr/bin/env python3
"""Does the TSDF session time depend on what the GPU did in the milliseconds before?  The bench sessions (19 scans back to
back, ~1 ms in all) repeated: per-scan time of every repetition, for the racing, merged and reproducible modes; then the
racing kernel by itself with the stream drained and (a) nothing, (b) a 50 ms sleep -- a sensor's period -- before each scan."""
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from harness.bench_tsdf import sensor_cases, session_scans  # noqa: E402


def main(scans=20, reps=12):
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
            lay = capi.TsdfLayer(ctx, vs, 16)
            for k in (0, scans - 1):
                lay.reserve(poses[k][4:7], reach)
            return lay
        res = {}
        gc.collect()
        gc.disable()
        for mode in ("racing", "merged", "reproducible"):
            cfg = capi.tsdf_config(**kw) if mode != "reproducible" else capi.tsdf_config(deterministic=1, **kw)
            per_rep = []
            time.sleep(0.2)           # the GPU idle before the first repetition, as before a bench block
            for r in range(reps):
                layer = new_layer()
                integ = capi.FastTsdfIntegrator(ctx, cfg, layer)
                call = integ.integrate_merged_device if mode == "merged" else integ.integrate_device
                call(poses[0], dev[0].data_ptr(), None, n_pts)
                ctx.synchronize()
                t0 = time.perf_counter()
                ctx.timer_start()
                for k in range(1, scans):
                    call(poses[k], dev[k].data_ptr(), None, n_pts)
                ms = ctx.timer_stop()
                wall = (time.perf_counter() - t0) * 1e3
                per_rep.append((round(ms / (scans - 1) * 1e3, 1), round(wall / (scans - 1) * 1e3, 1)))
                integ.destroy()
                layer.destroy()
            res[mode + "_us_per_scan_by_repetition(events, wall)"] = per_rep
        # the racing kernel by itself
        layer = new_layer()
        integ = capi.FastTsdfIntegrator(ctx, capi.tsdf_config(**kw), layer)
        for label, nap in (("drained", 0.0), ("after_50ms_idle", 0.05), ("drained_again", 0.0)):
            per = []
            for k in range(scans):
                ctx.synchronize()
                if nap:
                    time.sleep(nap)
                ctx.timer_start()
                integ.integrate_device(poses[k], dev[k].data_ptr(), None, n_pts)
                per.append(round(ctx.timer_stop() * 1e3, 1))
            res["racing_kernel_us_" + label] = per
        gc.enable()
        integ.destroy()
        layer.destroy()
        out[name] = res
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
