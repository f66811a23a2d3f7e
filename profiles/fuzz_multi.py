#!/usr/bin/env python3
"""Differential fuzzing of the sharded REG evaluation: random small pose graphs (3-7 submaps, random
constraint lists incl. repeated and mirrored pairs, self-registrations, constraints with no overlap at all, ESDF /
TSDF grids, voxel / isosurface points, a no-correspondence cost), random poses, sharded over a RANDOM number of
contexts (1-8, all on the one device) by a RANDOM placement (LPT, contiguous runs, or arbitrary -- so that shards
are empty, lopsided or interleaved):
  * vgx_reg_multi_evaluate_fused == the single batch's assembled buffer, BIT FOR BIT (np.uint64 views: the sign
    of a zero included), evaluation after evaluation and at a second set of poses;
  * vgx_reg_multi_evaluate_normal == the single batch's per-constraint blocks, bit for bit;
  * the one-process-per-rank route in miniature: every shard's vgx_reg_batch_scatter_normal array, the arrays
    summed as int64 words, vgx_reg_assembler_assemble -> the same bits again.
All-points constraints only: sampling constraints' draws depend on the placement by design.
    gpurun -- 'SEEDS=300 python profiles/fuzz_multi.py'"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
F = np.float32


def main():
    import torch
    from oracle import synth
    from tests import helpers as H
    from voxgraph_amd import capi
    capi.load()
    n_seeds, first = int(os.environ.get("SEEDS", "100")), int(os.environ.get("FIRST", "0"))
    ctxs = [capi.Context(0) for _ in range(8)]
    tally = {"graphs": 0, "constraints": 0, "empty_shards": 0, "placements": {"lpt": 0, "contiguous": 0, "arbitrary": 0}}
    for seed in range(first, first + n_seeds):
        rng = np.random.default_rng(seed)
        vps = int(rng.choice([8, 16]))
        vs = float(rng.choice([0.1, 0.2]))
        dims = tuple(int(x) for x in rng.integers(1, 3, 3))
        ext = np.array(dims) * vps * vs
        n_sub = int(rng.integers(3, 8))
        sms, true = [], []
        for k in range(n_sub):
            c = rng.uniform(0.2, 0.8, 3) * ext
            sdf = synth.sphere_ground_sdf(tuple(c), float(rng.uniform(0.25, 0.6) * ext.min()), float(rng.uniform(0.1, 0.4) * ext[2]))
            sm = synth.make_submap(sdf, vs, vps, (0, 0, 0), dims, trunc=3 * vs, esdf_max=10 * vs,
                                   drop_empty_blocks=bool(rng.integers(0, 2)))
            if len(sm.block_index) == 0:
                break
            sms.append(sm)
            true.append(np.r_[rng.uniform(-0.4, 0.4, 3) * ext * [1, 1, 0.2], rng.uniform(-0.5, 0.5)])
        if len(sms) < 3:
            continue
        n_sub = len(sms)
        n_con = int(rng.integers(1, 14))
        pairs = [(int(a), int(b)) for a, b in rng.integers(0, n_sub, (n_con, 2))]
        if rng.random() < 0.5:
            pairs += [(b, a) for a, b in pairs[:3]]                       # mirrored constraints (pose_graph.cpp:62-71)
        n_con = len(pairs)
        use_esdf = int(rng.integers(0, 2))
        iso = bool(rng.integers(0, 2))
        cfg = capi.default_config(registration_point_type=capi.POINTS_ISOSURFACE if iso else capi.POINTS_VOXELS,
                                  use_esdf_distance=use_esdf, no_correspondence_cost=float(rng.choice([0.0, 0.2])))
        n_ctx = int(rng.integers(1, 9))
        how = str(rng.choice(["lpt", "contiguous", "arbitrary"]))
        subs = []
        for k in range(n_ctx):                                            # every submap on every context (replicated)
            mine = []
            for i, sm in enumerate(sms):
                g = H.gpu_submap(capi, ctxs[k], sm, i)
                (g.extract_isosurface_points(1.0) if iso else g.extract_voxel_points(1.0, 0.3, bool(use_esdf)))
                mine.append(g)
            subs.append(mine)
        kind = capi.POINTS_ISOSURFACE if iso else capi.POINTS_VOXELS
        if min(g.num_points(kind) for g in subs[0]) == 0:                 # (a point set without points: nothing to shard)
            for mine in subs:
                for g in mine:
                    g.destroy()
            continue
        n_pts = [subs[0][a].num_points(kind) for a, _ in pairs]
        if how == "lpt":
            shard = capi.lpt_shards(n_pts, n_ctx)
        elif how == "contiguous":
            shard = capi.contiguous_shards(n_pts, n_ctx)
        else:
            shard = rng.integers(0, n_ctx, n_con).astype(np.int32)
        poses = np.array(true) + rng.normal(0, 1, (n_sub, 4)) * [vs, vs, 0.5 * vs, 0.03]
        poses2 = poses + rng.normal(0, 1, (n_sub, 4)) * [0.5 * vs, 0.5 * vs, 0.2 * vs, 0.01]
        made, cfs0, cfs = [], [], []
        try:
            cfs0 = [capi.RegistrationCostFunction(ctxs[0], subs[0][a], subs[0][b], cfg) for a, b in pairs]
            single = capi.RegistrationBatch(ctxs[0], cfs0, pairs)
            cfs = [capi.RegistrationCostFunction(ctxs[shard[c]], subs[shard[c]][a], subs[shard[c]][b], cfg)
                   for c, (a, b) in enumerate(pairs)]
            multi = capi.RegistrationMulti(ctxs[:n_ctx], cfs, pairs)
            made += [single, multi]
            asm = capi.RegistrationAssembler(ctxs[0], pairs)
            made.append(asm)
            size = capi.fused_size(n_sub, n_con)
            for ps in (poses, poses, poses2):
                _, normal0 = single.evaluate_normal(ps)
                buf = torch.full((size,), float("nan"), dtype=torch.float64, device="cuda:0")
                torch.cuda.synchronize()
                single.assemble(n_sub, buf.data_ptr(), zero_first=True)
                ctxs[0].synchronize()
                want = buf.cpu().numpy()
                fused, status = multi.evaluate_fused(ps)
                assert np.array_equal(fused.view(np.uint64), want.view(np.uint64)), ("fused buffer", float(np.abs(fused - want).max()))
                normal, _ = multi.evaluate_normal(ps)
                assert np.array_equal(normal.view(np.uint64), normal0.view(np.uint64)), "per-constraint blocks"
                # one process per rank, in miniature: scatter per shard, integer sum, one assembly
                total = np.zeros((n_con, capi.NORMAL_SIZE), np.uint64)
                for k in range(n_ctx):
                    mine = [c for c in range(n_con) if shard[c] == k]
                    bt = capi.RegistrationBatch(ctxs[k], [cfs[c] for c in mine], [pairs[c] for c in mine],
                                                global_index=mine, n_global=n_con)
                    bt.evaluate_normal(ps, to_host=False)
                    arr = torch.full((n_con, capi.NORMAL_SIZE), float("nan"), dtype=torch.float64, device="cuda:0")
                    torch.cuda.synchronize()
                    bt.scatter_normal(arr.data_ptr(), zero_first=True)
                    ctxs[k].synchronize()
                    total += arr.cpu().numpy().view(np.uint64)
                    bt.destroy()
                    tally["empty_shards"] += int(len(mine) == 0 and ps is poses2)
                blocks = torch.from_numpy(total.view(np.float64)).cuda()
                out = torch.full((size,), float("nan"), dtype=torch.float64, device="cuda:0")
                torch.cuda.synchronize()
                asm.assemble(blocks.data_ptr(), n_sub, out.data_ptr())
                ctxs[0].synchronize()
                assert np.array_equal(out.cpu().numpy().view(np.uint64), want.view(np.uint64)), "scatter / int64 sum / assemble"
        except AssertionError as e:
            print("MISMATCH", dict(seed=seed, n_sub=n_sub, pairs=pairs, n_ctx=n_ctx, how=how, shard=[int(x) for x in shard],
                                   iso=iso, use_esdf=use_esdf), str(e)[:300])
            return 1
        finally:
            for o in made:
                o.destroy()
            for o in cfs0 + cfs:
                o.destroy()
            for mine in subs:
                for g in mine:
                    g.destroy()
        tally["graphs"] += 1
        tally["constraints"] += n_con
        tally["placements"][how] += 1
    print("no mismatch:", tally)
    for c in ctxs:
        c.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
