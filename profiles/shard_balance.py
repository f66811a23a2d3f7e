#!/usr/bin/env python3
"""How would the bench graphs run on N = 2 / 4 / 8 GPUs?  (one GPU is enough to tell)

For config 3 (200 submaps @ 256^3, 1176 constraints) and config 5 (1000 submaps @ 128^3, 3807 constraints), two
placements of the constraint list -- vgx_lpt_shards (greedy longest-processing-time) and vgx_contiguous_shards
(consecutive runs of equal weight: the locality-aware alternative) -- both weighted with what a constraint moves
at the initial poses, 36 B x residuals + 45 B x live residuals (vgx_reg_batch_count_live_each, the round-3 tile
weights).  Every shard of every split is timed ALONE on this GPU (materialising pass and fused pass, HIP
events); an N-GPU pass takes as long as its slowest shard, so

    predicted_efficiency = single_batch_ms / (N x max_shard_ms)       (what SCALE would divide out, less the
                                                                       all-reduce of n x 360 B, latency bound)
    balance              = sum(shard_ms) / (N x max_shard_ms)

and per shard: the DISTINCT submaps its constraints touch -- what has to be resident on that GPU when submaps
are placed where they are referenced instead of replicated -- with the bytes that is (bricks + points).

    gpurun -- 'python profiles/shard_balance.py > gpurun_out/r04_shard_balance.json'
UNMEASURED ON MULTI-GPU HARDWARE: these are one-GPU timings of each shard.
"""
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from harness.bench_config5 import build_config5_graph, C5_BLOCK_DIMS, C5_BLOCK_MIN  # noqa: E402
from voxgraph_amd import capi  # noqa: E402


def study(name, ctx, torch, true_poses, poses, pairs, block_min, block_dims, splits=(2, 4, 8)):
    subs, n_points = [], []
    for k in range(len(true_poses)):
        sm = capi.Submap.synth_city(ctx, k, 0.2, 16, block_min, block_dims, 0.6, 2.0, 10.0, true_poses[k], 2)
        n_points.append(sm.extract_voxel_points(1.0, 0.3, True))
        sm.release_raw_layers()
        subs.append(sm)
    n_blocks = int(np.prod(block_dims))
    # resident bytes of a finished submap: apron bricks (17^3 f32 per block) + block table + 20 B per point
    submap_bytes = np.array([n_blocks * 17 ** 3 * 4 + n_blocks * 4 + 20 * n for n in n_points], np.int64)
    cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS)
    cfs = [capi.RegistrationCostFunction(ctx, subs[i], subs[j], cfg) for i, j in pairs]
    full = capi.RegistrationBatch(ctx, cfs, pairs)
    n_res = np.array([n_points[i] for i, _ in pairs], np.int64)
    live = full.count_live_each(poses)
    weights = 36 * n_res + 45 * live
    R = int(n_res.sum())
    res = torch.empty(R, dtype=torch.float32, device="cuda")
    jr = torch.empty((R, 4), dtype=torch.float32, device="cuda")
    je = torch.empty((R, 4), dtype=torch.float32, device="cuda")

    def time_batch(bt, fused, reps=8):
        f = (lambda: bt.evaluate_normal(poses, to_host=False)) if fused else \
            (lambda: bt.evaluate_points(poses, res.data_ptr(), jr.data_ptr(), je.data_ptr()))
        for _ in range(2):
            f()
        ctx.synchronize()
        ctx.timer_start()
        for _ in range(reps):
            f()
        return ctx.timer_stop() / reps
    single = {"points": time_batch(full, False), "fused": time_batch(full, True)}
    out = {"submaps": len(true_poses), "constraints": len(pairs), "residuals": R, "live_residuals": int(live.sum()),
           "weights": "36 B x residuals + 45 B x live residuals at the initial poses",
           "single_batch_ms": single, "replicated_bytes_per_gpu": int(submap_bytes.sum()),
           "allreduce_bytes_per_evaluation": int(len(pairs) * 45 * 8), "splits": {}}
    for N in splits:
        for pname, place in (("lpt", capi.lpt_shards), ("contiguous", capi.contiguous_shards)):
            shard_of = place(weights, N)
            tp, tf, touched, placed = [], [], [], []
            for r in range(N):
                mine = np.flatnonzero(shard_of == r).astype(np.int32)
                bt = capi.RegistrationBatch(ctx, [cfs[c] for c in mine], pairs[mine], global_index=mine, n_global=len(pairs))
                tp.append(time_batch(bt, False))
                tf.append(time_batch(bt, True))
                bt.destroy()
                ids = sorted({int(s) for c in mine for s in pairs[c]})
                touched.append(len(ids))
                placed.append(int(submap_bytes[ids].sum()))
            out["splits"][f"N{N}_{pname}"] = {
                "constraints_per_shard": np.bincount(shard_of, minlength=N).tolist(),
                "weight_balance": float(np.bincount(shard_of, weights=weights, minlength=N).mean()
                                        / np.bincount(shard_of, weights=weights, minlength=N).max()),
                "points_ms_per_shard": [round(t, 4) for t in tp], "fused_ms_per_shard": [round(t, 4) for t in tf],
                "points_balance": sum(tp) / (N * max(tp)), "fused_balance": sum(tf) / (N * max(tf)),
                "points_predicted_efficiency": single["points"] / (N * max(tp)),
                "fused_predicted_efficiency": single["fused"] / (N * max(tf)),
                "points_value_G_per_s_if_N_gpus": R / max(tp) / 1e6,
                "distinct_submaps_per_shard": touched,
                "distinct_submaps_fraction_max": max(touched) / len(true_poses),
                "placed_bytes_per_gpu_max": max(placed), "placed_bytes_all_gpus": int(sum(placed)),
                "replicated_bytes_all_gpus": int(submap_bytes.sum()) * N}
    full.destroy()
    for o in cfs + subs:
        o.destroy()
    del res, jr, je
    torch.cuda.empty_cache()
    return out


def main():
    import torch
    capi.load()
    ctx = capi.Context(0)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    a = types.SimpleNamespace(grid=[20, 10], block_dims=[16, 16, 16], block_min=[-8, -8, -4], voxel_size=0.2,
                              truncation=0.6, esdf_max=2.0, pose_sigma=0.3, yaw_sigma=0.05, seed=2)
    true3, poses3, pairs3 = bench.build_graph(a)
    out = {"what": __doc__.split("\n\n")[0], "unmeasured_on_multi_gpu_hardware": True,
           "config3": study("config3", ctx, torch, true3, poses3, pairs3, a.block_min, a.block_dims)}
    true5, pairs5, poses5, _, _ = build_config5_graph(25, 40)
    out["config5"] = study("config5", ctx, torch, true5, poses5, pairs5, C5_BLOCK_MIN, C5_BLOCK_DIMS)
    print(json.dumps(out, indent=1))
    ctx.close()


if __name__ == "__main__":
    main()
