#!/usr/bin/env python3
"""Differential fuzzing of the RACING TSDF mode by trace and replay (round 6): random integrator configurations x random
scan shapes (unorganised, organised with whole and ragged tiles) x sessions from a moving, turning sensor -- every scan run
by the event-logging instantiation of the shipped kernel, its log replayed through oracle/tsdf_replay.c (every set
exchange, every ray's decisions, every per-voxel fold: a legal interleaving of the sequential integrator's steps, bit for
bit).  Prints the first violation with its configuration, or a tally.
    gpurun -- 'SEEDS=200 python profiles/fuzz_replay.py'        BIG=1: larger scans, longer walks (the hand-on path)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
F = np.float32


def main():
    from oracle import pyoracle as orc
    from voxgraph_amd import capi
    from tests.test_tsdf_deterministic_gpu import _lidar_scan
    import torch
    capi.load()
    ctx = capi.Context(0)
    big = int(os.environ.get("BIG", "0"))
    n_seeds = int(os.environ.get("SEEDS", "100"))
    first = int(os.environ.get("FIRST", "7000"))
    tot = dict(sessions=0, scans=0, points=0, exchanges=0, updates=0, folds=0, overrun=0, max_overrun=0, skips=0, several_links=0,
               longest_fold=0, colour_writes=0)
    for seed in range(first, first + n_seeds):
        rng = np.random.default_rng(seed)
        vps = int(rng.choice([8, 16]))
        vs = float(rng.choice([0.05, 0.1, 0.2, 0.3]))
        kw = dict(default_truncation_distance=float(rng.uniform(1.5, 4) * vs),
                  max_ray_length_m=float(rng.uniform(15, 80 if big else 45) * vs),
                  min_ray_length_m=float(rng.uniform(0.3, 2) * vs),
                  voxel_carving_enabled=int(rng.integers(0, 2)), use_const_weight=int(rng.integers(0, 2)),
                  use_weight_dropoff=int(rng.integers(0, 2)),
                  use_sparsity_compensation_factor=int(rng.integers(0, 2)),
                  sparsity_compensation_factor=float(rng.uniform(1, 30)),
                  allow_clear=int(rng.integers(0, 2)), max_weight=float(rng.choice([3.0, 50.0, 10000.0])),
                  max_consecutive_ray_collisions=int(rng.choice([0, 1, 2, 2, 3, 5, 1 << 30])),
                  start_voxel_subsampling_factor=float(rng.choice([1.0, 2.0, 4.0])),
                  clear_checks_every_n_frames=int(rng.choice([1, 1, 2, 4])))
        ocfg, gcfg = orc.tsdf_config(**kw), capi.tsdf_config(**kw)
        layer = capi.TsdfLayer(ctx, vs, vps)
        integ = capi.FastTsdfIntegrator(ctx, gcfg, layer)
        integ.set_event_trace(48 << 20)
        room = ((-30 * vs, -24 * vs, -6 * vs), (32 * vs, 50 * vs, 14 * vs))
        n_scans = 6 if big else 4
        for k in range(n_scans):
            origin = (rng.uniform(-3, 3, 3) * vs).astype(F)
            if big:
                origin = (origin + np.array([6.0 * k, -4.0 * k, 0.3 * k]) * vs).astype(F)
            n_az = int(rng.integers(300, 1400)) if big else int(rng.integers(40, 500))
            n_el = int(rng.integers(16, 64)) if big else int(rng.integers(4, 40))
            pts = _lidar_scan(n_az, n_el, seed * 10 + k, room=room, origin=origin.astype(np.float64), el=0.5)
            mode = int(rng.integers(0, 3))          # 0: shuffled (unorganised); 1: organised, whole rows; 2: organised, another width
            width = 0
            if mode == 0:
                pts = pts[rng.permutation(len(pts))]
            elif mode == 1:
                width = n_az
            else:
                width = int(rng.choice([w for w in (n_az - 3, n_az // 2 + 1, 17, 100) if 0 < w <= len(pts)]))
                pts = pts[: len(pts) // width * width]
            pts[:2] = 0.0
            pts[2] = [np.nan, 1.0, 1.0]
            pts[rng.integers(0, len(pts), 5)] *= F(8.0)                       # far returns: clearing rays / dropped
            ang = rng.uniform(-3, 3)
            ax = rng.normal(0, 1, 3)
            ax /= np.linalg.norm(ax)
            T = np.r_[np.cos(ang / 2), np.sin(ang / 2) * ax, origin].astype(F)
            col = rng.integers(0, 256, (len(pts), 4)).astype(np.uint8) if rng.integers(0, 2) else None
            free = bool(k == n_scans - 1 and seed % 4 == 0)
            counted = bool(rng.integers(0, 2))
            integ.set_cloud_width(width)
            s0, o0, _ = integ.download_sets()
            l0 = layer.download()
            if counted or col is not None:
                n_upd = integ.integratePointCloud(T, pts, col, free)
            else:                                                             # the uncounted kernel, device-resident points
                d = torch.from_numpy(pts).cuda()
                torch.cuda.synchronize()
                integ.integrate_device(T, d.data_ptr(), None, len(pts), free)
                n_upd = None
            trace, lost = integ.read_event_trace()
            s1, o1, (off_s, off_o, _) = integ.download_sets()
            rep = orc.tsdf_replay_check(ocfg, vs, vps, T, pts, col, free, (off_s, off_o), (s0, o0), (s1, o1), l0, layer.download(), trace)
            bad = lost != 0 or rep["errors"] != 0 or layer.stats()[1] != 0 or (n_upd is not None and n_upd != rep["required_updates"])
            if bad:
                print("VIOLATION seed", seed, "scan", k, "lost", lost, "dropped", layer.stats()[1], "updates", n_upd, rep["required_updates"])
                print(" ", rep["first_error"])
                print("  config", kw, "vps", vps, "vs", vs, "width", width, "free", free, "counted", counted, "points", len(pts))
                return 1
            tot["scans"] += 1
            tot["points"] += len(pts)
            tot["exchanges"] += rep["observed_exchanges"]
            tot["updates"] += rep["required_updates"]
            tot["folds"] += rep["fold_events"]
            tot["overrun"] += rep["overrun_exchanges"]
            tot["max_overrun"] = max(tot["max_overrun"], rep["max_overrun"])
            tot["skips"] += rep["start_skips"]
            tot["several_links"] += rep["voxels_with_several_links"]
            tot["longest_fold"] = max(tot["longest_fold"], rep["longest_fold"])
            tot["colour_writes"] += rep["colour_writes"]
        tot["sessions"] += 1
        for o in (integ, layer):
            o.destroy()
    print("no violation:", tot)
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
