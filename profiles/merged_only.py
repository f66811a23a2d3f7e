#!/usr/bin/env python3
"""The sort-based TSDF paths by themselves on bench.py's two sensor shapes (for rocprofv3 --kernel-trace --stats):
INTEGRATOR=merged (default) | fast, DET=0|1 (fast: only DET=1 is sort-based), WHICH=lidar|rgbd|both."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from harness.bench_tsdf import _room_points
    from voxgraph_amd import capi
    capi.load()
    ctx = capi.Context(0)
    u, v = np.meshgrid((np.arange(640) - 319.5) / 525.0, (np.arange(480) - 239.5) / 525.0)
    d_rgbd = np.stack([np.ones_like(u), -u, -v], -1).reshape(-1, 3)
    d_rgbd /= np.linalg.norm(d_rgbd, axis=1, keepdims=True)
    az = np.linspace(-np.pi, np.pi, 1024, endpoint=False)
    el = np.deg2rad(np.linspace(-16.6, 16.6, 64))
    A, E = np.meshgrid(az, el)
    d_lidar = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], -1).reshape(-1, 3)
    which = os.environ.get("WHICH", "both")
    det = int(os.environ.get("DET", "0"))
    fast = os.environ.get("INTEGRATOR", "merged") == "fast"
    for name, dirs, vs, kw in (("rgbd", d_rgbd, 0.05, dict(default_truncation_distance=0.15, max_ray_length_m=5.0)),
                               ("lidar", d_lidar, 0.2, dict(default_truncation_distance=0.6, max_ray_length_m=16.0,
                                                            use_const_weight=1, use_weight_dropoff=1,
                                                            use_sparsity_compensation_factor=1,
                                                            sparsity_compensation_factor=20.0))):
        if which not in ("both", name):
            continue
        scans = 10
        poses, dev = [], []
        for k in range(scans):
            origin = np.array([-2.0 + 0.15 * k, 0.5 - 0.05 * k, 0.3 + 0.01 * k])
            yaw = 0.05 * k
            c, s_ = np.cos(yaw), np.sin(yaw)
            R = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1.0]])
            pts = (_room_points(dirs @ R.T, origin) @ R).astype(np.float32)
            poses.append(np.array([np.cos(yaw / 2), 0, 0, np.sin(yaw / 2), *origin], np.float32))
            dev.append(torch.from_numpy(pts).cuda())
        torch.cuda.synchronize()
        layer = capi.TsdfLayer(ctx, vs, 16)
        integ = capi.FastTsdfIntegrator(ctx, capi.tsdf_config(deterministic=det, **kw), layer)
        if os.environ.get("SPEC_DEPTH"):   # experiment: the reproducible mode's speculation depth (default 32)
            integ.set_speculation(int(os.environ["SPEC_DEPTH"]), int(os.environ.get("SPEC_THRESHOLD", str(8 << 20))))
        n = dev[0].shape[0]
        call = integ.integrate_device if fast else integ.integrate_merged_device
        call(poses[0], dev[0].data_ptr(), None, n)
        ctx.synchronize()
        t0 = time.perf_counter()
        for k in range(1, scans):
            call(poses[k], dev[k].data_ptr(), None, n)
        ctx.synchronize()
        print(name, "fast" if fast else "merged", "ms per scan", (time.perf_counter() - t0) * 1e3 / (scans - 1), "deterministic", det)
        integ.destroy(); layer.destroy()
    ctx.close()


if __name__ == "__main__":
    main()
