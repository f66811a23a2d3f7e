#!/usr/bin/env python3
"""One-off differential fuzzing of the REG path against oracle/reg_oracle.c: random small submap pairs (8 / 16
voxels per side, spheres + ground, blocks dropped at random), random poses (tiny, large, far away, exactly
aligned, yaw near +-pi), ESDF / TSDF grids, no-correspondence cost on / off, voxel and isosurface points:
  * drop-in vgx_reg_evaluate (f64): every residual and Jacobian entry EQUAL to the oracle's;
  * batched materialising pass (f32): EQUAL to the f32 rounding of the oracle's f64 values;
  * batched fused pass: 45 sums within 2e-6 of the sums of the oracle's rows (of each part's largest entry; 1e-6 is the
    usual worst, cancelling terms in a small constraint's J^T r have reached 1.3e-6), worst case reported;
  * the producers on the same submaps: extracted voxel points and isosurface vertices (random weights,
    min weight 1 / 6) EQUAL to the oracle's, order included.
    gpurun -- 'SEEDS=200 python profiles/fuzz_reg.py'"""
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
    from voxgraph_amd import capi
    capi.load()
    ctx = capi.Context(0)
    n_seeds, first = int(os.environ.get("SEEDS", "100")), int(os.environ.get("FIRST", "0"))
    done = 0
    worst, worst_mag, over = [0.0, 0.0, 0.0], [0.0, 0.0, 0.0], [0, 0, 0]
    for seed in range(first, first + n_seeds):
        rng = np.random.default_rng(seed)
        vps = int(rng.choice([8, 16]))
        vs = float(rng.choice([0.05, 0.1, 0.2]))
        dims = tuple(int(x) for x in rng.integers(1, 4, 3))
        ext = np.array(dims) * vps * vs
        sms = []
        for k in range(2):
            c = rng.uniform(0.2, 0.8, 3) * ext
            sdf = synth.sphere_ground_sdf(tuple(c), float(rng.uniform(0.2, 0.6) * ext.min()), float(rng.uniform(0.1, 0.4) * ext[2]))
            sm = synth.make_submap(sdf, vs, vps, tuple(int(x) for x in rng.integers(-2, 2, 3)), dims, trunc=3 * vs,
                                   esdf_max=10 * vs, drop_empty_blocks=bool(rng.integers(0, 2)))
            sms.append(sm)
        if min(len(s.block_index) for s in sms) == 0:
            continue
        gs = [H.gpu_submap(capi, ctx, sm, k) for k, sm in enumerate(sms)]
        use_esdf = int(rng.integers(0, 2))
        nc = float(rng.choice([0.0, 0.25]))
        kind = capi.POINTS_VOXELS
        n0 = gs[0].extract_voxel_points(1.0, float(rng.choice([0.3, 3 * vs])), bool(use_esdf))
        n1 = gs[1].extract_voxel_points(1.0, 0.3, bool(use_esdf))
        try:                                                                    # the two producers, bit for bit
            for k, (g, sm) in enumerate(zip(gs, sms)):
                md = 0.3 if k else None
                gx, gd, gw = g.download_points(capi.POINTS_VOXELS) if (n0 if k == 0 else n1) > 0 else (np.zeros((0, 3), F),) * 3
                if k == 1 and len(gw):
                    ox, od, ow = H.oracle_points(sm, use_esdf=bool(use_esdf), min_w=1.0, max_d=0.3)
                    assert np.array_equal(gx, ox) and np.array_equal(gd, od) and np.array_equal(gw, ow), "extracted voxel points"
                rw = np.where(sm.tsdf_weight > 0, rng.uniform(0.5, 12.0, sm.tsdf_weight.shape), 0).astype(F)
                g2 = capi.Submap(ctx, 7, sm.voxel_size, sm.vps, sm.block_index, sm.tsdf_distance, rw, sm.esdf_distance, sm.esdf_observed)
                min_w = float(rng.choice([1.0, 6.0]))
                ni = g2.extract_isosurface_points(min_w)
                ix, idist, iw = orc.isosurface_points(sm.voxel_size, sm.vps, sm.block_index, sm.tsdf_distance, rw, min_w)
                assert ni == len(iw), ("isosurface count", ni, len(iw))
                if ni:
                    jx, jd, jw = g2.download_points(capi.POINTS_ISOSURFACE)
                    assert np.array_equal(jx, ix) and np.array_equal(jd, idist) and np.array_equal(jw, iw), "isosurface points"
                g2.destroy()
        except AssertionError as e:
            print("MISMATCH producers", dict(seed=seed, vps=vps, vs=vs, dims=dims), str(e)[:300])
            return 1
        if n0 == 0:
            for g in gs:
                g.destroy()
            continue
        mode = seed % 6
        p0 = rng.normal(0, 1, 4) * [0.3, 0.3, 0.1, 0.2]
        p1 = p0 + rng.normal(0, 1, 4) * [2 * vs, 2 * vs, vs, 0.1]
        if mode == 1:
            p0 = np.zeros(4); p1 = np.zeros(4)                                  # exactly aligned grids
        elif mode == 2:
            p1 = p0 + np.array([vs * int(rng.integers(-3, 4)), vs * int(rng.integers(-3, 4)), 0.0, 0.0])  # whole voxels
        elif mode == 3:
            p1 = p0 + np.array([0, 0, 0, np.pi - 1e-4 * rng.uniform()])        # yaw near pi
        elif mode == 4:
            p1 = p0 + np.array([1e4, -3e3, 50.0, 1.0])                          # far away: nothing corresponds
        elif mode == 5:
            p1 = p0 + rng.normal(0, 1, 4) * [ext[0], ext[1], ext[2] / 2, 1.5]   # half out
        poses = np.array([p0, p1])
        cfg = capi.default_config(registration_point_type=kind, use_esdf_distance=use_esdf, no_correspondence_cost=nc)
        cf = capi.RegistrationCostFunction(ctx, gs[0], gs[1], cfg)
        n = cf.num_residuals()
        r, jo, je = np.zeros(n), np.zeros((n, 4)), np.zeros((n, 4))
        ok = cf.Evaluate([poses[0], poses[1]], r, [jo, je])
        layer = H.oracle_layer(sms[1], use_esdf=bool(use_esdf))
        xyz, dist, w = gs[0].download_points(kind)
        ok0, r0, jo0, je0 = orc.reg_evaluate(layer, xyz, dist, w, poses[0], poses[1], no_correspondence_cost=nc)
        what = dict(seed=seed, mode=mode, vps=vps, vs=vs, dims=dims, use_esdf=use_esdf, nc=nc, n=n)
        try:
            assert bool(ok) == bool(ok0), "return value"
            if ok0:
                assert np.array_equal(r, r0) and np.array_equal(jo, jo0) and np.array_equal(je, je0), "drop-in rows"
            cf_rev = capi.RegistrationCostFunction(ctx, gs[1], gs[0], cfg) if n1 > 0 else None
            cfs = [cf] + ([cf_rev] if cf_rev else [])
            pairs = [(0, 1)] + ([(1, 0)] if cf_rev else [])
            batch = capi.RegistrationBatch(ctx, cfs, pairs)
            R = batch.num_residuals()
            tr = torch.full((R,), float("nan"), dtype=torch.float32, device="cuda:0")
            tjo = torch.full((R, 4), float("nan"), dtype=torch.float32, device="cuda:0")
            tje = torch.full((R, 4), float("nan"), dtype=torch.float32, device="cuda:0")
            torch.cuda.synchronize()
            st = batch.evaluate_points(poses, tr.data_ptr(), tjo.data_ptr(), tje.data_ptr())
            ctx.synchronize()
            if ok0:
                assert st[0] == 0, ("status", st)
                assert np.array_equal(tr[:n].cpu().numpy(), r0.astype(F)), "batch residual rows"
                assert np.array_equal(tjo[:n].cpu().numpy(), jo0.astype(F)), "batch jac_ref rows"
                assert np.array_equal(tje[:n].cpu().numpy(), je0.astype(F)), "batch jac_read rows"
                st2, normal = batch.evaluate_normal(poses)
                st3, cost = batch.evaluate_cost(poses)        # (round 6) the cost-only pass: the full pass's cost bit for bit
                assert np.array_equal(cost.view(np.uint64), normal[:, 0].copy().view(np.uint64)), ("cost-only", cost, normal[:, 0])
                J = np.concatenate([jo0, je0], axis=1)
                want = np.r_[float(r0 @ r0), J.T @ r0, (J.T @ J)[np.triu_indices(8)]]
                # what the sums are MADE of: sum |J_i r_i| per entry -- an f32 pass cannot do better than a few ulp of THAT when
                # the terms cancel (seed 300290: J^T r at 1.27e-6 of its largest entry, 2e-7 of the terms' magnitude)
                mag = np.r_[float(r0 @ r0), np.abs(J).T @ np.abs(r0), (np.abs(J).T @ np.abs(J))[np.triu_indices(8)]]
                for gi, (lo, hi) in enumerate(((0, 1), (1, 9), (9, 45))):
                    scale = np.abs(want[lo:hi]).max()
                    if scale > 0:
                        err = np.abs(normal[0][lo:hi] - want[lo:hi]).max() / scale
                        err_mag = (np.abs(normal[0][lo:hi] - want[lo:hi]) / np.maximum(mag[lo:hi], 1e-300)).max()
                        worst[gi] = max(worst[gi], err)
                        worst_mag[gi] = max(worst_mag[gi], err_mag)
                        over[gi] += err > 1e-6
                        assert err <= 2e-6 or err_mag <= 5e-7, ("fused", lo, err, err_mag)
                    else:
                        assert np.abs(normal[0][lo:hi]).max() == 0, ("fused zero part", lo)
            batch.destroy()
            if cf_rev:
                cf_rev.destroy()
        except AssertionError as e:
            print("MISMATCH", what, str(e)[:400])
            return 1
        done += 1
        cf.destroy()
        for g in gs:
            g.destroy()
    print("no mismatch in", done, "random constraint evaluations (drop-in f64 rows, batched f32 rows, fused sums); fused sums' worst "
          "error relative to the largest entry of (cost, J^T r, J^T J): %.2e %.2e %.2e (above 1e-6: %d %d %d constraints), relative "
          "to the magnitude of their terms: %.2e %.2e %.2e" % (*worst, *over, *worst_mag))
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
