#!/usr/bin/env python3
"""One-off differential fuzzing of the SAMPLING mode (registration_cost_function.cpp:113-122,
weighted_sampler_inl.h:18-28): random ratios (0.01-2.5), random non-uniform weights (some zero), shared /
private engines, Morton-sorted or uploaded order, drop-in and batched evaluations interleaved at random --
every evaluation must consume the engines' std::mt19937 streams exactly as the oracle does and return the
oracle's rows (drop-in f64: equal; batched f32: equal to the f32 rounding).
    gpurun -- 'SEEDS=150 python profiles/fuzz_sampling.py'"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
F = np.float32


def main():
    import torch
    from oracle import pyoracle as orc
    from oracle import synth
    from tests import helpers as H
    from tests.test_ref_pin import sequential_cumsum
    from voxgraph_amd import capi
    capi.load()
    ctx = capi.Context(0)
    n_seeds, first = int(os.environ.get("SEEDS", "100")), int(os.environ.get("FIRST", "0"))
    evals = 0
    for seed in range(first, first + n_seeds):
        rng = np.random.default_rng(seed)
        vs = 0.1
        sdf = synth.sphere_ground_sdf(tuple(rng.uniform(1.0, 2.2, 3)), float(rng.uniform(0.6, 1.1)), 0.35)
        sms, pts, gs = [], [], []
        for k in range(2):
            sm = synth.make_submap(sdf, vs, 16, (0, 0, 0), (2, 2, 2), trunc=0.3, esdf_max=1.0, pose=(0.3 * k, 0.1 * k, 0, 0.1 * k),
                                   drop_empty_blocks=True)
            xyz, dist, w = H.oracle_points(sm)
            keep = rng.permutation(len(w))[: int(rng.integers(50, max(51, min(len(w), 3000))))]
            keep.sort()
            xyz, dist, w = xyz[keep], dist[keep], (w[keep] * rng.uniform(0.0, 1.0, len(keep)) * (rng.uniform(size=len(keep)) > 0.1)).astype(F)
            if w.sum() <= 0:
                w[0] = 1.0
            g = H.gpu_submap(capi, ctx, sm, k)
            g.set_points(capi.POINTS_VOXELS, xyz, dist, w, capi.POINTS_SORT_MORTON if rng.integers(0, 2) else capi.POINTS_KEEP_ORDER)
            sms.append(sm); pts.append((xyz, dist, w)); gs.append(g)
        ratio = float(rng.choice([0.01, 0.05, 0.3, 1.0, 2.5]))
        private = int(rng.integers(0, 2)) * int(rng.integers(1, 1000))
        cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS, sampling_ratio=ratio, sampler_seed=private)
        pairs = [(0, 1), (1, 0), (0, 1)]
        cfs = [capi.RegistrationCostFunction(ctx, gs[a], gs[b], cfg) for a, b in pairs]
        if min(cf.num_residuals() for cf in cfs) == 0:
            for o in cfs + gs:
                o.destroy()
            continue
        batch = capi.RegistrationBatch(ctx, cfs, pairs)
        ro = batch.row_offsets()
        layers = [H.oracle_layer(sm) for sm in sms]
        cums = [sequential_cumsum(p[2]) for p in pts]
        if private:
            engines = [orc.Mt19937(private) for _ in cfs]           # one per cost function
            eng_of = lambda c: engines[c]
        else:
            shared = {a: orc.Mt19937(5489) for a in (0, 1)}          # one per reference point set
            eng_of = lambda c: shared[pairs[c][0]]
        poses = np.array([[0.02, -0.01, 0.0, 0.01], [0.33, 0.08, 0.02, 0.12]])

        def oracle_rows(c):
            a, b = pairs[c]
            xyz, dist, w = pts[a]
            n = cfs[c].num_residuals()
            assert n == int(F(ratio) * F(len(w))), ("num_residuals", n, len(w), ratio)
            idx = np.array([eng_of(c).weighted_draw(cums[a]) for _ in range(n)], np.int64)
            ok, r0, jo0, je0 = orc.reg_evaluate(layers[b], xyz, dist, w, poses[a], poses[b], sample_idx=idx)
            return ok, r0, jo0, je0

        try:
            for step in range(5):
                if rng.integers(0, 2):                               # a batched evaluation
                    R = batch.num_residuals()
                    tr = torch.full((R,), float("nan"), dtype=torch.float32, device="cuda:0")
                    tjo = torch.full((R, 4), float("nan"), dtype=torch.float32, device="cuda:0")
                    tje = torch.full((R, 4), float("nan"), dtype=torch.float32, device="cuda:0")
                    torch.cuda.synchronize()
                    batch.evaluate_points(poses, tr.data_ptr(), tjo.data_ptr(), tje.data_ptr())
                    ctx.synchronize()
                    for c in range(len(pairs)):
                        ok, r0, jo0, je0 = oracle_rows(c)
                        s = slice(ro[c], ro[c + 1])
                        assert np.array_equal(tr[s].cpu().numpy(), r0.astype(F)), ("batch residuals", step, c)
                        assert np.array_equal(tjo[s].cpu().numpy(), jo0.astype(F)) and np.array_equal(tje[s].cpu().numpy(), je0.astype(F)), ("batch jacobians", step, c)
                else:                                                # a drop-in Evaluate of one cost function
                    c = int(rng.integers(0, len(pairs)))
                    n = cfs[c].num_residuals()
                    r, j1, j2 = np.zeros(n), np.zeros((n, 4)), np.zeros((n, 4))
                    a, b = pairs[c]
                    ok = cfs[c].Evaluate([poses[a], poses[b]], r, [j1, j2])
                    ok0, r0, jo0, je0 = oracle_rows(c)
                    assert bool(ok) == bool(ok0), ("return", step, c)
                    assert np.array_equal(r, r0) and np.array_equal(j1, jo0) and np.array_equal(j2, je0), ("drop-in", step, c)
                evals += 1
        except AssertionError as e:
            print("MISMATCH seed", seed, "ratio", ratio, "private", private, str(e)[:300])
            return 1
        for o in [batch] + cfs + gs:
            o.destroy()
    print("no mismatch in", evals, "evaluations (batched and drop-in, interleaved)")
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
