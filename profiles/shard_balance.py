#!/usr/bin/env python3
"""How well do the LPT shards of config 3 balance on N GPUs?  (one GPU is enough to tell)

Every shard of an N-way split is timed alone on this GPU (materialising pass and fused pass, HIP
events); the N-GPU pass takes as long as the slowest shard, so  sum(t) / (N * max(t))  is the balance
part of the scaling efficiency (the all-reduce of the fused buffer is not in it).  Two weightings of
vgx_lpt_shards are compared: the residual count (round 1) and the bytes a constraint moves at the
initial poses, 36 B x residuals + 45 B x live residuals (vgx_reg_batch_count_live_each).

    gpurun -- 'python profiles/shard_balance.py > gpurun_out/shard_balance.json'
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from voxgraph_amd import capi  # noqa: E402


def main():
    import torch
    capi.load()

    class A:
        pass
    a = A()
    a.grid, a.block_dims, a.block_min, a.voxel_size = [20, 10], [16, 16, 16], [-8, -8, -4], 0.2
    a.truncation, a.esdf_max, a.pose_sigma, a.yaw_sigma, a.seed = 0.6, 2.0, 0.3, 0.05, 2
    true_poses, poses, pairs = bench.build_graph(a)
    ctx = capi.Context(0)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    subs, n_points = [], []
    for k in range(len(true_poses)):
        sm = capi.Submap.synth_city(ctx, k, 0.2, 16, a.block_min, a.block_dims, 0.6, 2.0, 10.0, true_poses[k], 2)
        n_points.append(sm.extract_voxel_points(1.0, 0.3, True))
        sm.release_raw_layers()
        subs.append(sm)
    cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS)
    cfs = [capi.RegistrationCostFunction(ctx, subs[i], subs[j], cfg) for i, j in pairs]
    full = capi.RegistrationBatch(ctx, cfs, pairs)
    n_res = np.array([n_points[i] for i, _ in pairs], np.int64)
    live = full.count_live_each(poses)
    R = int(n_res.sum())
    res = torch.empty(R, dtype=torch.float32, device="cuda")
    jr = torch.empty((R, 4), dtype=torch.float32, device="cuda")
    je = torch.empty((R, 4), dtype=torch.float32, device="cuda")
    weightings = {"residuals": n_res, "bytes_36N_45live": 36 * n_res + 45 * live}

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
    out = {"constraints": len(pairs), "residuals": R, "live_residuals": int(live.sum()),
           "single_batch_ms": {"points": time_batch(full, False), "fused": time_batch(full, True)}, "splits": {}}
    for N in (2, 4, 8):
        for name, w in weightings.items():
            shards = bench.lpt_shards(w, N)
            tp, tf = [], []
            for mine in shards:
                mine = np.array(mine, np.int32)
                bt = capi.RegistrationBatch(ctx, [cfs[c] for c in mine], pairs[mine], global_index=mine, n_global=len(pairs))
                tp.append(time_batch(bt, False))
                tf.append(time_batch(bt, True))
                bt.destroy()
            out["splits"][f"N{N}_{name}"] = {
                "points_ms_per_shard": [round(t, 4) for t in tp], "fused_ms_per_shard": [round(t, 4) for t in tf],
                "points_balance": sum(tp) / (N * max(tp)), "fused_balance": sum(tf) / (N * max(tf)),
                "points_value_G_per_s_if_N_gpus": R / max(tp) / 1e6,
                "residuals_per_shard": [int(n_res[m].sum()) for m in shards]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
