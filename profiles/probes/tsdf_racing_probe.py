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
        width = int(os.environ.get("VGX_PROBE_ORGANISED", "0")) and (640 if name.startswith("rgbd") else 1024)
        if width:
            integ.set_cloud_width(width)
        integ.integrate_device(poses[0], dev[0].data_ptr(), None, n_pts)
        ctx.synchronize()
        ctx.timer_start()
        for k in range(1, scans):
            integ.integrate_device(poses[k], dev[k].data_ptr(), None, n_pts)
        b2b = ctx.timer_stop() / (scans - 1)
        layer2 = new_layer()
        integ.setLayer(layer2)
        per, host_us, grow = [], [], []
        import time as _t
        for k in range(scans):
            ctx.synchronize()
            ctx.timer_start()
            h0 = _t.perf_counter()
            integ.integrate_device(poses[k], dev[k].data_ptr(), None, n_pts)
            host_us.append((_t.perf_counter() - h0) * 1e6)
            per.append(ctx.timer_stop())
            grow.append(layer2.growths())
        layer3 = new_layer()
        integ.setLayer(layer3)
        stats, traces = [], []
        wgs = (n_pts + 255) // 256 if not width else ((width + 15) // 16) * ((n_pts // width + 15) // 16)
        v1 = os.environ.get("VGX_TSDF_KERNEL") == "v1"
        for k in range(scans):
            u = integ.integrate_device(poses[k], dev[k].data_ptr(), None, n_pts, count=True)
            stats.append(dict(integ.walk_stats(), updates=u))
            if not v1 and k >= 1:
                t = integ.read_trace(wgs)
                has_walk = t[:, 2] > 0
                traces.append({"start_spread_us": float(t[:, 0].max()), "span_us": float(t[:, 3].max()),
                               "phase1_us_mean": float((t[:, 1] - t[:, 0]).mean()), "phase1_us_max": float((t[:, 1] - t[:, 0]).max()),
                               "walk_us_mean": float((t[has_walk, 2] - t[has_walk, 1]).mean()) if has_walk.any() else 0.0,
                               "walk_us_max": float((t[has_walk, 2] - t[has_walk, 1]).max()) if has_walk.any() else 0.0,
                               "flush_us_mean": float((t[has_walk, 3] - t[has_walk, 2]).mean()) if has_walk.any() else 0.0,
                               "flush_us_max": float((t[has_walk, 3] - t[has_walk, 2]).max()) if has_walk.any() else 0.0,
                               "wg_us_mean": float((t[:, 3] - t[:, 0]).mean()), "wg_us_max": float((t[:, 3] - t[:, 0]).max()),
                               "workgroups_with_rays": int(has_walk.sum()),
                               "rays_max": float(t[:, 4].max()), "rays_total": float(t[:, 4].sum()), "rounds_max": float(t[:, 5].max()),
                               "folds_max": float(t[:, 6].max()), "retry_chain_max": float(t[:, 7].max())})
                if k == 5:
                    worst = np.argsort(t[:, 3] - t[:, 0])[::-1][:6]
                    slowest = [{"wg": int(w_), "us": [round(float(x), 2) for x in t[w_, :4]], "rays": int(t[w_, 4]), "rounds": int(t[w_, 5]),
                                "folds": int(t[w_, 6]), "retry_chain": int(t[w_, 7])} for w_ in worst]
        # the floor of the measurement itself: a one-point scan timed the same way
        one = []
        for k in range(1, 8):
            ctx.synchronize()
            ctx.timer_start()
            integ.integrate_device(poses[k], dev[k].data_ptr(), None, 1)
            one.append(ctx.timer_stop())
        nb, dropped = layer.stats()
        out[name] = {"cloud_width": width, "kernel_us_first_scan": per[0] * 1e3, "kernel_us": float(np.mean(per[1:])) * 1e3, "kernel_us_median": float(np.median(per[1:])) * 1e3,
                     "kernel_us_each": [round(x * 1e3, 1) for x in per], "host_call_us_each": [round(x, 1) for x in host_us],
                     "growths_each": grow,
                     "kernel_us_min": float(np.min(per[1:])) * 1e3, "kernel_us_max": float(np.max(per[1:])) * 1e3,
                     "back_to_back_us": b2b * 1e3, "blocks": nb, "dropped": dropped,
                     "one_point_scan_us": float(np.median(one)) * 1e3,
                     "trace": {k_: float(np.mean([t_[k_] for t_ in traces])) for k_ in traces[0]} if traces else None,
                     "slowest_workgroups_scan5": slowest if traces else None,
                     "first_scan": stats[0],
                     "per_scan": {k_: float(np.mean([s_[k_] for s_ in stats[1:]])) for k_ in stats[0]}}
        for o in (integ, layer, layer2, layer3):
            o.destroy()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
