#!/usr/bin/env python3
"""The racing TSDF kernel by itself on the two bench sessions: per-scan kernel time (HIP events, stream drained before each
launch), back-to-back time, and what the rays did (vgx_tsdf_integrator_walk_stats).  The kernel is chosen by the
environment (VGX_TSDF_KERNEL=v1: the one-thread-per-point kernel of rounds 1-4), so an A/B is two processes:
    VGX_TSDF_KERNEL=v1 python profiles/probes/tsdf_racing_probe.py ; python profiles/probes/tsdf_racing_probe.py
Also checks the layer of every session against the other modes' order-independent facts: update count > 0, nothing dropped."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from harness.bench_tsdf import sensor_cases, session_scans  # noqa: E402


def main(scans=20):
    import torch
    from voxgraph_amd import capi
    capi.load()
    ctx = capi.Context(0)
    out = {"kernel": os.environ.get("VGX_TSDF_KERNEL", "v2 (cooperative)")}
    for name, (dirs, vs, kw, _, _) in sensor_cases().items():
        poses, clouds = session_scans(dirs, scans)
        n_pts = clouds[0].shape[0]
        reach = kw["max_ray_length_m"] + kw["default_truncation_distance"] + 2 * vs

        def new_layer():
            lay = capi.TsdfLayer(ctx, vs, 16)
            for k in (0, scans - 1):
                lay.reserve(poses[k][4:7], reach)
            return lay
        dev = [torch.from_numpy(c_).cuda() for c_ in clouds]
        layer = new_layer()
        integ = capi.FastTsdfIntegrator(ctx, capi.tsdf_config(**kw), layer)
        integ.integrate_device(poses[0], dev[0].data_ptr(), None, n_pts)
        ctx.synchronize()
        ctx.timer_start()
        for k in range(1, scans):
            integ.integrate_device(poses[k], dev[k].data_ptr(), None, n_pts)
        b2b = ctx.timer_stop() / (scans - 1)
        layer2 = new_layer()
        integ.setLayer(layer2)
        per = []
        for k in range(scans):
            ctx.synchronize()
            ctx.timer_start()
            integ.integrate_device(poses[k], dev[k].data_ptr(), None, n_pts)
            per.append(ctx.timer_stop())
        layer3 = new_layer()
        integ.setLayer(layer3)
        stats = []
        for k in range(scans):
            u = integ.integrate_device(poses[k], dev[k].data_ptr(), None, n_pts, count=True)
            stats.append(dict(integ.walk_stats(), updates=u))
        nb, dropped = layer.stats()
        out[name] = {"kernel_us_first_scan": per[0] * 1e3, "kernel_us": float(np.mean(per[1:])) * 1e3,
                     "kernel_us_min": float(np.min(per[1:])) * 1e3, "kernel_us_max": float(np.max(per[1:])) * 1e3,
                     "back_to_back_us": b2b * 1e3, "blocks": nb, "dropped": dropped,
                     "first_scan": stats[0],
                     "per_scan": {k_: float(np.mean([s_[k_] for s_ in stats[1:]])) for k_ in stats[0]}}
        for o in (integ, layer, layer2, layer3):
            o.destroy()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
