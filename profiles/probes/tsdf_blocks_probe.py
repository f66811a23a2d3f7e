#!/usr/bin/env python3
"""Racing LiDAR kernel, stream drained before each launch: a fresh layer (pass 1) against the same scans into the same layer
again (passes 2, 3), with the layer's block count after every scan -- is the 42 vs 22 us of tsdf_warm_probe.py the blocks a
scan has to allocate?"""
import gc
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
        lay = new_layer()
        integ = capi.FastTsdfIntegrator(ctx, capi.tsdf_config(**kw), lay)
        gc.collect()
        gc.disable()
        res = {}
        keep = []
        for p in ("1 fresh layer, fresh integrator", "2 same again", "3 same again", "4 counted", "5 same layer, NEW integrator",
                  "6 same again", "7 NEW layer, same integrator", "8 same again", "9 same again",
                  "4b counted, NEW integrator (age 0-19)", "4c counted, same again (age 20-39)"):
            if p.startswith("4b"):
                keep.append(integ)
                integ = capi.FastTsdfIntegrator(ctx, capi.tsdf_config(**kw), lay)
            if p.startswith("5"):
                keep.append(integ)
                integ = capi.FastTsdfIntegrator(ctx, capi.tsdf_config(**kw), lay)
            if p.startswith("7"):
                keep.append(lay)
                lay = new_layer()
                integ.setLayer(lay)
            per, blocks, walks, traces = [], [], [], []
            for k in range(scans):
                ctx.synchronize()
                ctx.timer_start()
                u = integ.integrate_device(poses[k], dev[k].data_ptr(), None, n_pts, count=p.startswith("4"))
                per.append(round(ctx.timer_stop() * 1e3, 1))
                blocks.append(int(lay.stats()[0]))
                if p.startswith("4"):
                    walks.append(dict(integ.walk_stats(), updates=u))
                    if k >= 1:
                        t = integ.read_trace((n_pts + 255) // 256)
                        hw = t[:, 2] > 0
                        traces.append({"phase1_mean": float((t[:, 1] - t[:, 0]).mean()), "phase1_max": float((t[:, 1] - t[:, 0]).max()),
                                       "walk_mean": float((t[hw, 2] - t[hw, 1]).mean()), "walk_max": float((t[hw, 2] - t[hw, 1]).max()),
                                       "flush_mean": float((t[hw, 3] - t[hw, 2]).mean()), "flush_max": float((t[hw, 3] - t[hw, 2]).max()),
                                       "wg_mean": float((t[:, 3] - t[:, 0]).mean()), "wg_max": float((t[:, 3] - t[:, 0]).max()),
                                       "span": float(t[:, 3].max()), "rays": float(t[:, 4].sum()), "rays_max": float(t[:, 4].max()),
                                       "rounds_sum": float(t[:, 5].sum()), "rounds_max": float(t[:, 5].max()),
                                       "folds_max": float(t[:, 6].max()), "retry_chain_max": float(t[:, 7].max())})
            res[f"pass {p}"] = {"us": per, "median_us": float(np.median(per[1:])), "blocks_after": blocks}
            if walks:
                res[f"pass {p}"]["per_scan"] = {k_: float(np.mean([w[k_] for w in walks[1:]])) for k_ in walks[0]}
                res[f"pass {p}"]["trace"] = {k_: round(float(np.mean([t_[k_] for t_ in traces])), 2) for k_ in traces[0]}
        gc.enable()
        out[name] = res
        integ.destroy()
        lay.destroy()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
