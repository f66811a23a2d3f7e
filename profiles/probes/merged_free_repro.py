#!/usr/bin/env python3
"""fuzz_tsdf.py BIG=1 seed 5022 (merged integrator, reproducible mode, a freespace scan at scan 5): where do the GPU
layer and the oracle's differ?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
F = np.float32


def main():
    from oracle import pyoracle as orc
    from voxgraph_amd import capi
    from tests.test_tsdf_deterministic_gpu import _lidar_scan
    capi.load()
    ctx = capi.Context(0)
    seed = int(os.environ.get("SEED", "5022"))
    det = int(os.environ.get("DET", "1"))
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
    for k_, v_ in os.environ.items():
        if k_.startswith("KW_"):
            kw[k_[3:]] = type(kw[k_[3:]])(float(v_))
    order = int(rng.integers(0, 2))
    order = int(os.environ.get("ORDER", order))
    print("config", kw, "vps", vps, "vs", vs, "order", order)
    ol, gl = orc.TsdfLayer(vs, vps), capi.TsdfLayer(ctx, vs, vps)
    oi = orc.FastTsdfIntegrator(orc.tsdf_config(integration_order=order + 1, **kw), ol)
    gi = capi.FastTsdfIntegrator(ctx, capi.tsdf_config(deterministic=det, integration_order=order, **kw), gl)
    room = ((-30 * vs, -24 * vs, -6 * vs), (32 * vs, 50 * vs, 14 * vs))
    srng = np.random.default_rng(seed * 7 + 1)
    only = os.environ.get("ONLY")   # integrate only this scan (fresh layers)
    for k in range(8):
        origin = (srng.uniform(-3, 3, 3) * vs).astype(F)
        origin = (origin + np.array([10.0 * k, -6.0 * k, 0.5 * k]) * vs).astype(F)
        size = (int(srng.integers(400, 1100)), int(srng.integers(16, 40)))
        pts = _lidar_scan(size[0], size[1], seed * 10 + k, room=room, origin=origin.astype(np.float64), el=0.5)
        pts = pts[srng.permutation(len(pts))]
        pts[:3] = 0.0
        pts[3] = [np.nan, 1.0, 1.0]
        ang = srng.uniform(-3, 3)
        ax = srng.normal(0, 1, 3); ax /= np.linalg.norm(ax)
        T = np.r_[np.cos(ang / 2), np.sin(ang / 2) * ax, origin].astype(F)
        col = srng.integers(0, 256, (len(pts), 4)).astype(np.uint8)
        free = bool(k % 3 == 2 and seed % 3 == 0)
        if only is not None and k != int(only):
            continue
        a = oi.integratePointCloudMerged(T, pts, col, free)
        b = gi.integratePointCloudMerged(T, pts, col, free)
        obi, od, ow, oc = ol.download()
        gbi, gd, gw, gc = gl.download()
        o_ord = np.lexsort(obi.T[::-1]); g_ord = np.lexsort(gbi.T[::-1])
        same_blocks = obi.shape == gbi.shape and np.array_equal(obi[o_ord], gbi[g_ord])
        rd = np.linalg.norm(pts[np.isfinite(pts).all(1)], axis=1)
        print(f"scan {k} free {free} points {len(pts)} r min/max {rd.min():.3f} {rd.max():.3f} updates oracle {a} gpu {b} blocks {len(obi)} / {len(gbi)} "
              f"same set {same_blocks} stats {gl.stats()} growths {gl.growths()}")
        if a != b or not same_blocks or not np.array_equal(ow[o_ord].view(np.uint32), gw[g_ord].view(np.uint32)):
            if same_blocks:
                dw = ow[o_ord] != gw[g_ord]
                blocks = np.flatnonzero(dw.reshape(len(obi), -1).any(1))
                print(" voxels with different weight:", int(dw.sum()), "in", len(blocks), "blocks")
                bi = obi[o_ord][blocks]
                centre = (bi + 0.5) * vs * vps
                dist = np.linalg.norm(centre - origin, axis=1)
                print(" block indices (first 10):", bi[:10].tolist(), "distance of block centres from the sensor:", np.round(dist[:10], 2).tolist())
                w_o, w_g = ow[o_ord][dw], gw[g_ord][dw]
                print(" oracle / gpu weights (first 10):", w_o[:10].tolist(), w_g[:10].tolist())
                print(" weight differences: oracle higher", int((w_o > w_g).sum()), "gpu higher", int((w_g > w_o).sum()))
            else:
                so = {tuple(x) for x in obi.tolist()}; sg = {tuple(x) for x in gbi.tolist()}
                print(" only oracle:", sorted(so - sg)[:10], len(so - sg), " only gpu:", sorted(sg - so)[:10], len(sg - so))
            break
    ctx.close()


if __name__ == "__main__":
    main()
