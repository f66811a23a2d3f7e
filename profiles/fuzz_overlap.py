#!/usr/bin/env python3
"""One-off differential fuzzing of vgx_find_overlapping_pairs against oracle/overlap_oracle.py (the numpy
restatement of voxgraph_submap.cpp:245-321, pose_graph_interface.cpp:109-147): random clusters of 4-9
submaps at random poses (touching, nested, far apart, rotated), pair lists must be equal.
    gpurun -- 'SEEDS=200 python profiles/fuzz_overlap.py'"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from oracle import overlap_oracle as ovl
    from oracle import pyoracle as orc
    from oracle import synth
    from tests import helpers as H
    from voxgraph_amd import capi
    capi.load()
    ctx = capi.Context(0)
    n_seeds, first = int(os.environ.get("SEEDS", "100")), int(os.environ.get("FIRST", "0"))
    total_pairs = done = 0
    for seed in range(first, first + n_seeds):
        rng = np.random.default_rng(seed)
        vps = int(rng.choice([8, 16]))
        vs = float(rng.choice([0.1, 0.2]))
        dims = (2, 2, 1) if vps == 16 else (3, 3, 2)
        ext = np.array(dims) * vps * vs
        sdf = synth.union_sdf(synth.sphere_ground_sdf(tuple(rng.uniform(0.3, 0.7, 3) * ext), float(0.3 * ext.min()), float(0.2 * ext[2])),
                              synth.sphere_sdf(tuple(rng.uniform(0, 1, 3) * ext), float(0.2 * ext.min())))
        subs, gs, poses = [], [], []
        for i in range(int(rng.integers(4, 10))):
            p = np.r_[rng.normal(0, 1, 2) * ext[0] * rng.choice([0.3, 1.0, 4.0]), rng.normal(0, 0.2) * ext[2], rng.uniform(-np.pi, np.pi)]
            sm = synth.make_submap(sdf, vs, vps, (0, 0, 0), dims, trunc=3 * vs, pose=tuple(p), esdf_max=5 * vs,
                                   drop_empty_blocks=True)
            if sm.n_blocks == 0:
                continue
            g = H.gpu_submap(capi, ctx, sm, i)
            nv, ni = g.extract_voxel_points(), g.extract_isosurface_points()
            if nv == 0 or ni == 0:
                g.destroy()
                continue
            vx, _, _ = H.oracle_points(sm)
            ix, _, _ = orc.isosurface_points(sm.voxel_size, vps, sm.block_index, sm.tsdf_distance, sm.tsdf_weight)
            subs.append(dict(voxel_size=sm.voxel_size, vps=vps, block_index=sm.block_index, voxel_xyz=vx, iso_xyz=ix))
            gs.append(g)
            poses.append(p + np.r_[rng.normal(0, 0.05, 3), rng.normal(0, 0.02)])
        if len(gs) >= 2:
            poses = np.array(poses)
            got = capi.find_overlapping_pairs(ctx, gs, poses)
            want = ovl.overlapping_pairs(subs, poses)
            if got != want:
                print("MISMATCH seed", seed, "got", got, "want", want)
                return 1
            total_pairs += len(got)
            done += 1
        for g in gs:
            g.destroy()
    print("no mismatch:", done, "clusters,", total_pairs, "overlapping pairs in all")
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
