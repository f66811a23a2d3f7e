#!/usr/bin/env python3
"""One-off differential fuzzing of the reproducible TSDF mode against oracle/tsdf_oracle.c: many more random
integrator configurations than tests/test_tsdf_deterministic_gpu.py runs, both integrators (fast / merged),
both modes of the merged one; the fast integrator's reproducible mode with a random speculation depth / threshold
(vgx_tsdf_integrator_set_speculation: 1-32 steps, from "always cut" to the default) and, every other session, the
same number of points in every scan, so that the extension marks kept between scans and the warm second attempt
are in play.  Prints the first mismatch with its configuration, or a tally.
    gpurun -- 'SEEDS=60 python profiles/fuzz_tsdf.py'"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
F = np.float32


def main():
    from oracle import pyoracle as orc
    from voxgraph_amd import capi
    from tests.test_tsdf_deterministic_gpu import _lidar_scan, _assert_layers_identical
    capi.load()
    ctx = capi.Context(0)
    big = int(os.environ.get("BIG", "0"))      # BIG=1: eight larger scans per session, the sensor moving 2-3 m a scan
    n_seeds = int(os.environ.get("SEEDS", "40"))
    first = int(os.environ.get("FIRST", "1000"))
    tally = {"fast": 0, "merged": 0, "merged racing": 0}
    for seed in range(first, first + n_seeds):
        rng = np.random.default_rng(seed)
        vps = int(rng.choice([8, 16]))
        vs = float(rng.choice([0.05, 0.1, 0.2, 0.3]))
        kw = dict(default_truncation_distance=float(rng.uniform(1.5, 4) * vs),
                  max_ray_length_m=float(rng.uniform(15, 45) * vs),
                  min_ray_length_m=float(rng.uniform(0.3, 2) * vs),
                  voxel_carving_enabled=int(rng.integers(0, 2)), use_const_weight=int(rng.integers(0, 2)),
                  use_weight_dropoff=int(rng.integers(0, 2)),
                  use_sparsity_compensation_factor=int(rng.integers(0, 2)),
                  sparsity_compensation_factor=float(rng.uniform(1, 30)),
                  allow_clear=int(rng.integers(0, 2)), max_weight=float(rng.choice([3.0, 50.0, 10000.0])),
                  max_consecutive_ray_collisions=int(rng.integers(0, 4)),
                  start_voxel_subsampling_factor=float(rng.choice([1.0, 2.0, 4.0])),
                  enable_anti_grazing=int(rng.integers(0, 2)))
        order = int(rng.integers(0, 2))          # integration_order_mode: 0 "mixed", 1 "sorted" (the oracle counts 1 / 2)
        spec = (int(rng.choice([1, 2, 3, 5, 9, 32])), int(rng.choice([0, 0, 500, 8 << 20])))
        same_size = bool(rng.integers(0, 2))
        for kind in ("fast", "merged", "merged racing"):
            det = 0 if kind == "merged racing" else 1
            ol, gl = orc.TsdfLayer(vs, vps), capi.TsdfLayer(ctx, vs, vps)
            oi = orc.FastTsdfIntegrator(orc.tsdf_config(integration_order=order + 1, **kw), ol)
            gi = capi.FastTsdfIntegrator(ctx, capi.tsdf_config(deterministic=det, integration_order=order, **kw), gl)
            if kind == "fast":
                gi.set_speculation(*spec)
            room = ((-30 * vs, -24 * vs, -6 * vs), (32 * vs, 50 * vs, 14 * vs))
            srng = np.random.default_rng(seed * 7 + 1)
            size = None
            for k in range(8 if big else 3):
                origin = (srng.uniform(-3, 3, 3) * vs).astype(F)
                if big:
                    origin = (origin + np.array([10.0 * k, -6.0 * k, 0.5 * k]) * vs).astype(F)   # the layer grows, the table is re-boxed
                if size is None or not same_size:
                    size = (int(srng.integers(400, 1100)) if big else int(srng.integers(40, 400)),
                            int(srng.integers(16, 40)) if big else int(srng.integers(4, 30)))
                pts = _lidar_scan(size[0], size[1], seed * 10 + k, room=room, origin=origin.astype(np.float64), el=0.5)
                pts = pts[srng.permutation(len(pts))]
                pts[:3] = 0.0
                pts[3] = [np.nan, 1.0, 1.0]
                ang = srng.uniform(-3, 3)
                ax = srng.normal(0, 1, 3); ax /= np.linalg.norm(ax)
                T = np.r_[np.cos(ang / 2), np.sin(ang / 2) * ax, origin].astype(F)
                col = srng.integers(0, 256, (len(pts), 4)).astype(np.uint8)
                free = bool(k % 3 == 2 and seed % 3 == 0)
                try:
                    if kind == "fast":
                        a = oi.integratePointCloud(T, pts, col, free)
                        b = gi.integratePointCloud(T, pts, col, free)
                    else:
                        a = oi.integratePointCloudMerged(T, pts, col, free)
                        b = gi.integratePointCloudMerged(T, pts, col, free)
                    assert a == b, ("updates", a, b)
                    if det:
                        _assert_layers_identical(ol, gl, f"{kind} seed {seed} scan {k}")
                    else:                                   # values exact, block order is arrival order
                        obi, od, ow, oc = ol.download()
                        gbi, gd, gw, gc = gl.download()
                        o_ord = np.lexsort(obi.T[::-1]); g_ord = np.lexsort(gbi.T[::-1])
                        assert np.array_equal(obi[o_ord], gbi[g_ord]), "block set"
                        assert np.array_equal(od[o_ord].view(np.uint32), gd[g_ord].view(np.uint32)), "distance"
                        assert np.array_equal(ow[o_ord].view(np.uint32), gw[g_ord].view(np.uint32)), "weight"
                        assert np.array_equal(oc[o_ord], gc[g_ord]), "colour"
                except AssertionError as e:
                    print("MISMATCH", kind, "seed", seed, "scan", k, "free", free, kw, "vps", vps, "vs", vs, "integration_order", order,
                          "speculation", spec, "same_size", same_size)
                    print(str(e)[:600])
                    return 1
            tally[kind] += 1
            for o in (gi, gl):
                o.destroy()
    print("no mismatch:", tally, "configurations x", 8 if big else 3, "scans each")
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
