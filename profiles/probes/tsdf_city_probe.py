#!/usr/bin/env python3
"""The racing TSDF kernel on the config-2 session's scans (64 x 1024 beams sphere-traced against the analytic city, 16 m
rays, 0.2 m voxels: LONG rays and many cast rays per workgroup -- the regime opposite to the room sessions of
tsdf_racing_probe.py): median kernel time per scan (HIP events, stream drained before) and one counted scan's statistics.
The kernel is chosen by the environment (VGX_TSDF_KERNEL=v1) as in tsdf_racing_probe.py."""
import gc
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main(n_submaps=2, scans_per_submap=30):
    import torch
    from harness import pipeline
    from voxgraph_amd import capi
    capi.load()
    ctx = capi.Context(0)
    n_az, n_el, vs = 1024, 64, 0.2
    cfg = capi.voxgraph_tsdf_config()
    sensor_poses = pipeline.session_sensor_poses(n_submaps, scans_per_submap, None)
    pts = torch.empty((n_az * n_el, 3), dtype=torch.float32, device="cuda")
    organised = int(os.environ.get("VGX_PROBE_ORGANISED", "0"))
    per, stats, trace = [], None, None
    integ = None
    gc.disable()
    for m in range(n_submaps):
        first = m * scans_per_submap
        P = sensor_poses[first].copy()
        P[2] = 0.0
        layer = capi.TsdfLayer(ctx, vs, 16)
        for j in (first, first + scans_per_submap - 1):
            layer.reserve(pipeline._inv_compose(P, sensor_poses[j])[4:7], 16.0 + 0.6 + 2 * vs)
        if integ is None:
            integ = capi.FastTsdfIntegrator(ctx, cfg, layer)
            if organised:
                integ.set_cloud_width(n_az)
        else:
            integ.setLayer(layer)
        for j in range(first, first + scans_per_submap):
            capi.synth_city_scan(ctx, sensor_poses[j], n_az, n_el, np.deg2rad(33.2), 40.0, 2, pts.data_ptr())
            ctx.synchronize()
            T = pipeline._inv_compose(P, sensor_poses[j])
            if m == 1 and j == first + 10:
                u = integ.integrate_device(T, pts.data_ptr(), None, n_az * n_el, count=True)
                stats = dict(integ.walk_stats(), updates=u)
                if os.environ.get("VGX_TSDF_KERNEL") != "v1":
                    t = integ.read_trace(4096)
                    has = t[:, 2] > 0
                    trace = {"rays_total": float(t[:, 4].sum()), "rays_max": float(t[:, 4].max()), "rounds_max": float(t[:, 5].max()),
                             "folds_total": float(t[:, 6].sum()), "span_us": float(t[:, 3].max()),
                             "phase1_us_mean": float((t[:, 1] - t[:, 0]).mean()),
                             "walk_us_mean": float((t[has, 2] - t[has, 1]).mean()), "walk_us_max": float((t[has, 2] - t[has, 1]).max()),
                             "flush_us_mean": float((t[has, 3] - t[has, 2]).mean()), "flush_us_max": float((t[has, 3] - t[has, 2]).max())}
                continue
            ctx.timer_start()
            integ.integrate_device(T, pts.data_ptr(), None, n_az * n_el)
            per.append(ctx.timer_stop())
    gc.enable()
    print(json.dumps({"kernel": os.environ.get("VGX_TSDF_KERNEL", "coop"), "organised": organised,
                      "kernel_us_median": float(np.median(per)) * 1e3, "kernel_us_min": float(np.min(per)) * 1e3,
                      "kernel_us_mean": float(np.mean(per)) * 1e3, "kernel_us_max": float(np.max(per)) * 1e3, "stats": stats, "trace": trace}))


if __name__ == "__main__":
    main()
