#!/usr/bin/env python3
"""One-off differential check of the SAMPLING mode at sizes where the draw's bucket table, the device
mt19937 streams (hundreds of twists per engine) and the sampled fused / materialising tiles all matter:
128^3 city submaps, isosurface or voxel points (10^4-10^5 per submap), sampling_ratio 0.05 / 0.5 / 1.3, engines
shared per reference submap, three evaluations in a row; every materialised f32 row against the oracle fed
with the oracle's own std::mt19937 draws.
    gpurun -- 'SEEDS=6 python profiles/fuzz_sampling_large.py'"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
F = np.float32


def main():
    import torch
    from oracle import pyoracle as orc
    from tests.test_ref_pin import sequential_cumsum
    from voxgraph_amd import capi
    capi.load()
    ctx = capi.Context(0)
    n_seeds, first = int(os.environ.get("SEEDS", "4")), int(os.environ.get("FIRST", "0"))
    vs, vps = 0.2, 16
    bmin, bdim = (-4, -4, -2), (8, 8, 8)
    rows = evals = 0
    worst = [0.0, 0.0, 0.0]
    for seed in range(first, first + n_seeds):
        rng = np.random.default_rng(seed)
        n_sub = 3
        true = np.c_[rng.uniform(-8, 8, (n_sub, 2)), rng.uniform(-0.3, 0.3, n_sub), rng.uniform(-0.3, 0.3, n_sub)]
        kind = capi.POINTS_ISOSURFACE if seed % 2 == 0 else capi.POINTS_VOXELS
        ratio = float([0.05, 0.5, 1.3][seed % 3])
        subs, layers, pts, cums = [], [], [], []
        for k in range(n_sub):
            sm = capi.Submap.synth_city(ctx, k, vs, vps, bmin, bdim, 0.6, 2.0, 10.0, true[k], seed % 5)
            n = sm.extract_isosurface_points(1.0) if kind == capi.POINTS_ISOSURFACE else sm.extract_voxel_points(1.0, 0.3, True)
            td, tw, ed, eo = sm.download_layers(vps)
            layers.append(orc.Layer(vs, vps, sm.block_index(), ed, eo))
            p = sm.download_points(kind) if n else None
            pts.append(p)
            cums.append(sequential_cumsum(p[2]) if n else None)
            subs.append(sm)
        pairs = [(a, b) for a in range(n_sub) for b in range(n_sub) if a != b and pts[a] is not None]
        cfg = capi.default_config(registration_point_type=kind, sampling_ratio=ratio)
        cfs = [capi.RegistrationCostFunction(ctx, subs[a], subs[b], cfg) for a, b in pairs]
        batch = capi.RegistrationBatch(ctx, cfs, pairs)
        ro = batch.row_offsets()
        R = batch.num_residuals()
        engines = {a: orc.Mt19937(5489) for a in range(n_sub)}
        for ev in range(3):
            poses = true + rng.normal(0, 1, (n_sub, 4)) * [0.3, 0.3, 0.05, 0.05]
            fused_first = ev == 1                                  # the fused pass consumes a whole evaluation's draws too
            if fused_first:
                _, normal = batch.evaluate_normal(poses)
            else:
                tr = torch.full((R,), float("nan"), dtype=torch.float32, device="cuda:0")
                tjo = torch.full((R, 4), float("nan"), dtype=torch.float32, device="cuda:0")
                tje = torch.full((R, 4), float("nan"), dtype=torch.float32, device="cuda:0")
                torch.cuda.synchronize()
                batch.evaluate_points(poses, tr.data_ptr(), tjo.data_ptr(), tje.data_ptr())
                ctx.synchronize()
                r, jo, je = tr.cpu().numpy(), tjo.cpu().numpy(), tje.cpu().numpy()
            for c, (a, b) in enumerate(pairs):
                xyz, dist, w = pts[a]
                n = cfs[c].num_residuals()
                idx = np.array([engines[a].weighted_draw(cums[a]) for _ in range(n)], np.int64)
                ok, r0, jo0, je0 = orc.reg_evaluate(layers[b], xyz, dist, w, poses[a], poses[b], sample_idx=idx)
                s = slice(ro[c], ro[c + 1])
                if fused_first:
                    J = np.concatenate([jo0, je0], axis=1)
                    want = np.r_[float(r0 @ r0), J.T @ r0, (J.T @ J)[np.triu_indices(8)]]
                    good = True
                    for part, (lo, hi) in enumerate(((0, 1), (1, 9), (9, 45))):
                        scale = np.abs(want[lo:hi]).max()
                        if scale > 0:
                            err = float(np.abs(normal[c][lo:hi] - want[lo:hi]).max() / scale)
                            worst[part] = max(worst[part], err)
                            good = good and err <= 2e-6
                else:
                    good = ok and np.array_equal(r[s], r0.astype(F)) and np.array_equal(jo[s], jo0.astype(F)) and np.array_equal(je[s], je0.astype(F))
                    rows += n
                if not good:
                    print("MISMATCH seed", seed, "evaluation", ev, "constraint", (a, b), "ratio", ratio, "kind", kind, "n", n, "points", len(w))
                    return 1
            evals += 1
        for o in [batch] + cfs + subs:
            o.destroy()
    print("no mismatch:", evals, "batched evaluations,", rows, "sampled rows compared exactly; fused sums' worst relative error "
          "(cost, J^T r, J^T J): %.2e %.2e %.2e" % tuple(worst))
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
