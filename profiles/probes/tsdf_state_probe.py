#!/usr/bin/env python3
"""The racing LiDAR kernel by itself runs at ~22 us in some stretches of a process and at ~40 us in others
(profiles/probes/tsdf_warm_probe.py).  Same workload every series here (a fresh layer + integrator, the same 20 scans, stream
drained before each launch), different things done BEFORE the series; the card's clocks (sysfs, found by PCI bus id) are
read between scans, outside the timed interval."""
import gc
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from harness.bench_tsdf import sensor_cases, session_scans  # noqa: E402
from harness import box_state  # noqa: E402


def main(scans=20):
    import torch
    from voxgraph_amd import capi
    capi.load()
    ctx = capi.Context(0)
    dev_dir = box_state._device_dir(0)
    hw = box_state._hwmon(dev_dir) if dev_dir else None
    dpm = sorted(glob.glob(os.path.join(dev_dir, "pp_dpm_*"))) if dev_dir else []

    def clocks():
        row = {}
        for p in dpm:
            a, _ = box_state._active_level(box_state._read(p))
            row[os.path.basename(p)[7:]] = a
        if hw:
            for k, f in (("sclk_in", "freq1_input"), ("mclk_in", "freq2_input"), ("power", "power1_average")):
                v = box_state._read(os.path.join(hw, f))
                row[k] = round(float(v) / 1e6) if v and v.isdigit() else None
        return row
    name = [k for k in sensor_cases() if k.startswith("lidar")][0]
    dirs, vs, kw, _, _ = sensor_cases()[name]
    poses, clouds = session_scans(dirs, scans)
    n_pts = clouds[0].shape[0]
    reach = kw["max_ray_length_m"] + kw["default_truncation_distance"] + 2 * vs
    dev = [torch.from_numpy(c_).cuda() for c_ in clouds]
    big_a = torch.empty(1 << 30, dtype=torch.float32, device="cuda")
    big_b = torch.empty(1 << 30, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()

    def series(sample, between=None):
        lay = capi.TsdfLayer(ctx, vs, 16)
        for k in (0, scans - 1):
            lay.reserve(poses[k][4:7], reach)
        integ = capi.FastTsdfIntegrator(ctx, capi.tsdf_config(**kw), lay)
        per, rows = [], []
        for k in range(scans):
            ctx.synchronize()
            if between:
                between()
            if sample:
                rows.append(clocks())
            ctx.timer_start()
            integ.integrate_device(poses[k], dev[k].data_ptr(), None, n_pts)
            per.append(round(ctx.timer_stop() * 1e3, 1))
        integ.destroy()
        lay.destroy()
        return per, rows

    def load(seconds):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            big_b.copy_(big_a)
            torch.cuda.synchronize()

    def short_copy():
        big_b[:1 << 24].copy_(big_a[:1 << 24])   # 64 MiB each way, ~25 us
        torch.cuda.synchronize()
    out = {"dpm_files": [os.path.basename(p) for p in dpm], "clocks_at_start": clocks()}
    gc.collect()
    gc.disable()
    steps = [("first", None, None), ("again", None, None), ("sampled", None, "sample"),
             ("after_1s_of_copies", lambda: load(1.0), None), ("again_2", None, None),
             ("after_2s_idle", lambda: time.sleep(2.0), None), ("again_3", None, None),
             ("a_64MiB_copy_before_each_scan", None, short_copy), ("sampled_2", None, "sample"),
             ("1ms_sleep_before_each_scan", None, lambda: time.sleep(0.001)),
             ("20ms_sleep_before_each_scan", None, lambda: time.sleep(0.02)), ("again_4", None, None)]
    for label, before, per_scan in steps:
        if before:
            before()
        c0 = clocks()
        per, rows = series(per_scan == "sample", per_scan if callable(per_scan) else None)
        entry = {"us": per, "median_us": float(np.median(per[1:])), "clocks_before": c0}
        if rows:
            keys = [k for k in rows[0] if len({str(r[k]) for r in rows}) > 1]
            entry["clocks_that_changed"] = {k: [r[k] for r in rows] for k in keys}
        out[label] = entry
    gc.enable()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
