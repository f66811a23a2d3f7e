"""bench.py's config-5 block: BASELINE configs[4], 1000 submaps @ 128^3, loop closures, two-stage solve."""
import os
import sys
import time

import numpy as np

from harness.bench_common import HBM_PEAK_GBS, shards  # noqa: F401

C5_VOXEL_SIZE, C5_BLOCK_DIMS, C5_BLOCK_MIN = 0.2, (8, 8, 8), (-4, -4, -2)      # 128^3 voxels, 25.6 m cubes


def build_config5_graph(n_lanes, per_lane):
    """BASELINE configs[4]'s pose graph: n_lanes x per_lane submaps on a serpentine trajectory (50 % overlap along
    a lane, 25 % across), registration constraints between consecutive submaps and across neighbouring lanes,
    odometry edges with accumulated drift, 20 injected loop closures (seed 4).
    Returns (true poses, registration pairs, odometry poses, edges, number of loop closures)."""
    from harness import lm
    n = n_lanes * per_lane
    rng = np.random.default_rng(4)                                 # seed 4
    dx, dy = 12.8, 19.2                                             # 50 % overlap along a lane, 25 % across

    def idx(lane, q):                                               # path index of x-position q in a lane
        return lane * per_lane + (q if lane % 2 == 0 else per_lane - 1 - q)
    true = np.zeros((n, 4))
    for lane in range(n_lanes):
        for q in range(per_lane):
            true[idx(lane, q)] = [q * dx, lane * dy, 0.0, rng.uniform(-0.1, 0.1)]
    pairs = [(i, i + 1) for i in range(n - 1)]
    for lane in range(n_lanes - 1):
        for q in range(per_lane):
            for dq in (-1, 0, 1):
                if 0 <= q + dq < per_lane:
                    a, b = idx(lane, q), idx(lane + 1, q + dq)
                    if abs(a - b) > 1:
                        pairs.append((min(a, b), max(a, b)))
    pairs = np.array(sorted(set(pairs)), np.int32)

    def between(pa, pb):
        c, s_ = np.cos(pa[3]), np.sin(pa[3])
        d = pb[:3] - pa[:3]
        return np.array([c * d[0] + s_ * d[1], -s_ * d[0] + c * d[1], d[2], lm.normalize_angle(pb[3] - pa[3])])

    def compose(pose, delta):
        c, s_ = np.cos(pose[3]), np.sin(pose[3])
        return np.array([pose[0] + c * delta[0] - s_ * delta[1], pose[1] + s_ * delta[0] + c * delta[1],
                         pose[2] + delta[2], lm.normalize_angle(pose[3] + delta[3])])
    # odometry: good in z and yaw (the yaml's information 2500), drifting in x, y.  The per-step noise
    # is sized so that neighbours across lanes (40-80 steps apart along the path) start within the
    # registration basin (a few voxels), as they do when voxgraph optimises after every new submap
    sig = np.array([0.01, 0.01, 0.001, 5e-5])
    info_odo = [1.0, 1.0, 2500.0, 2500.0]                            # voxgraph_mapper.yaml:41-47
    info_lc = [100.0, 100.0, 2500.0, 2500.0]                         # not in the yaml (template is zero): 0.1 m
    poses0 = true[:1].copy()
    edges = []
    for k in range(n - 1):
        delta = between(true[k], true[k + 1]) + rng.normal(0, sig)
        edges.append(lm.RelativePoseEdge(k, k + 1, delta[:3], delta[3], info_odo))
        poses0 = np.vstack([poses0, compose(poses0[k], delta)])
    n_lc = 20
    for j in range(n_lc):
        lane = 1 + (j * (n_lanes - 1)) // n_lc
        q = (7 * j + 3) % per_lane
        a, b = idx(lane - 1, q), idx(lane, q)
        # a loop closure's yaw error acts over the whole lane behind it (1 mrad over 500 m = 0.5 m), so
        # a usable one is accurate to a fraction of that
        delta = between(true[a], true[b]) + rng.normal(0, [0.03, 0.03, 0.005, 1e-4])
        edges.append(lm.RelativePoseEdge(a, b, delta[:3], delta[3], info_lc))
    return true, pairs, poses0, edges, n_lc


def config5_bench(capi, ctx, torch, dist, use_dist, rank, world, args):
    """BASELINE configs[4]: 1000 submaps @ 128^3 on a loop (serpentine) trajectory, odometry edges
    with accumulated drift, 20 injected loop-closure relative-pose edges
    (PoseGraphInterface::addLoopClosureMeasurement, pose_graph_interface.cpp:68-92) and the
    reference's two-stage optimisation (PoseGraphInterface::optimize, :177-198: loop closures are
    new, so first optimise WITHOUT the registration constraints, then with all of them).
    Constraints pair-sharded over the ranks; one all-reduce per solver evaluation."""
    from harness import lm
    from harness.backends import GpuBackend
    n_lanes, per_lane = args.config5_grid
    n = n_lanes * per_lane
    vs, dims, bmin = C5_VOXEL_SIZE, C5_BLOCK_DIMS, C5_BLOCK_MIN
    true, pairs, poses0, edges, n_lc = build_config5_graph(n_lanes, per_lane)

    t0 = time.perf_counter()
    submaps, n_points = [], []
    for k in range(n):
        sm = capi.Submap.synth_city(ctx, k, vs, 16, bmin, dims, args.truncation, args.esdf_max, 10.0,
                                    true[k], args.seed)
        n_points.append(sm.extract_voxel_points(1.0, 0.3, True))
        sm.release_raw_layers()
        submaps.append(sm)
    ctx.synchronize()
    setup_s = time.perf_counter() - t0
    cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS)
    mine = shards([n_points[a] for a, _ in pairs], world)[rank]
    cfs = [capi.RegistrationCostFunction(ctx, submaps[pairs[c][0]], submaps[pairs[c][1]], cfg) for c in mine]
    batch = capi.RegistrationBatch(ctx, cfs, pairs[mine], global_index=mine, n_global=len(pairs))
    backend = GpuBackend(capi, ctx, batch, n, dist if use_dist else None, node_pair_global=pairs)

    def barrier():
        if use_dist:
            dist.barrier()

    def rmse(p):
        return float(np.sqrt(((p[:, :3] - true[:, :3]) ** 2).sum(1).mean()))

    def rmse_aligned(p):
        """absolute trajectory error after the best rigid alignment (yaw + translation) of the whole
        estimate onto the truth: what is left once the gauge -- which submap 0 alone holds, through the
        few constraints it takes part in -- is taken out"""
        a, b = p[:, :2] - p[:, :2].mean(0), true[:, :2] - true[:, :2].mean(0)
        H = a.T @ b
        th = np.arctan2(H[0, 1] - H[1, 0], H[0, 0] + H[1, 1])
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        dxy = a @ R.T - b
        dz = (p[:, 2] - p[:, 2].mean()) - (true[:, 2] - true[:, 2].mean())
        return float(np.sqrt((dxy ** 2).sum(1).mean() + (dz ** 2).mean()))
    kw = dict(parameter_tolerance=1e-10, max_seconds=1e9)            # Ceres-default function_tolerance decides
    if os.environ.get("VGX_C5_DEBUG"):
        def parts(p):
            reg = float(backend(p)[0]) * 0.5
            tot = lm.Problem(backend, n, pairs, edges).evaluate_reduced(p)[0]
            return {"registration": reg, "edges": tot - reg, "rmse": rmse(p)}
        print("C5 at truth", parts(true), file=sys.stderr)
        print("C5 at odometry", parts(poses0), file=sys.stderr)
        xt, st = lm.solve(lm.Problem(backend, n, pairs, edges), true, **kw)
        print("C5 from truth ->", parts(xt), st["iterations"], st["termination"], file=sys.stderr)
        xo, so = lm.solve(lm.Problem(backend, n, pairs, edges), poses0, **kw)
        print("C5 from odometry (no stage 1) ->", parts(xo), so["iterations"], so["termination"], file=sys.stderr)
        err = np.linalg.norm((xo - true)[:, :2], axis=1)
        print("C5 error by lane", [round(float(err[l * per_lane:(l + 1) * per_lane].mean()), 3) for l in range(n_lanes)], file=sys.stderr)
        print("C5 aligned rmse: odometry", rmse_aligned(poses0), "from odometry ->", rmse_aligned(xo), file=sys.stderr)
        reg_only = lm.Problem(backend, n, pairs, [])
        xr, sr = lm.solve(reg_only, poses0, **kw)
        print("C5 registration only from odometry ->", rmse(xr), sr["iterations"], sr["termination"], file=sys.stderr)
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):                 # the two-stage solve, step by step
            lm.optimize_two_stage(backend, n, pairs, edges, poses0, True, verbose=True, **kw)
    lm.optimize_two_stage(backend, n, pairs, edges, poses0, True, **kw)      # untimed warm-up
    torch.cuda.synchronize()
    barrier()
    s0 = time.perf_counter()
    x, summaries = lm.optimize_two_stage(backend, n, pairs, edges, poses0, True, **kw)
    torch.cuda.synchronize()
    barrier()
    sdt = torch.tensor([time.perf_counter() - s0], dtype=torch.float64, device="cuda")
    # stage 1 alone, for the intermediate error
    x1, _ = lm.solve(lm.Problem(lm.zero_registration_backend(n, len(pairs)), n, pairs, edges), poses0, **kw)
    # one fused evaluation of every registration constraint at the initial guess, timed by itself
    for _ in range(2):
        backend(poses0)
    torch.cuda.synchronize()
    barrier()
    e0 = time.perf_counter()
    for _ in range(10):
        backend(poses0)
    torch.cuda.synchronize()
    barrier()
    edt = torch.tensor([(time.perf_counter() - e0) / 10], dtype=torch.float64, device="cuda")
    rs = torch.tensor([float(batch.num_residuals())], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(sdt, op=dist.ReduceOp.MAX)
        dist.all_reduce(edt, op=dist.ReduceOp.MAX)
        dist.all_reduce(rs, op=dist.ReduceOp.SUM)
    it_1e3, t_1e3 = next(((int(it_), float(t_)) for (it_, c_), t_ in zip(summaries[1]["cost_history"],
                                                                         summaries[1]["seconds_history"])
                          if c_ <= summaries[1]["final_cost"] * (1 + 1e-3)), (None, summaries[1]["seconds"]))
    out = {"workload": f"configs[4]: {n} submaps @ 128^3 (0.2 m) on a serpentine loop trajectory "
                       f"({n_lanes} lanes x {per_lane}), {len(pairs)} registration constraints (kVoxels, all points), "
                       f"{n - 1} odometry edges with accumulated drift, {n_lc} injected loop-closure edges; "
                       "two-stage optimisation (pose_graph_interface.cpp:177-198)",
           "submaps": n, "registration_constraints": int(len(pairs)), "loop_closures": n_lc,
           "residuals_per_evaluation": float(rs.item()),
           "solve_ms": float(sdt.item()) * 1e3,
           "solve_gpu_evaluation_ms": sum(s_["backend_seconds"] for s_ in summaries) * 1e3,
           "solve_host_linear_algebra_ms": sum(s_["host_linear_algebra_seconds"] for s_ in summaries) * 1e3,
           "stage1_without_registration": {k: summaries[0][k] for k in ("iterations", "evaluations", "termination")},
           "stage2_all_constraints": {k: summaries[1][k] for k in ("iterations", "evaluations", "termination",
                                                                  "initial_cost", "final_cost")},
           # Past the first few iterations stage 2 walks at random among the kinks of the trilinear field
           # (steps of 1e-5 m, gain ratios between -50 and +100: VGX_C5_DEBUG=1 prints them), so WHEN
           # function_tolerance or the iteration cap ends it depends on the last bits of the sums: the
           # iteration count, and with it solve_ms, is not a property of the kernels.  These two are:
           "stage2_ms_per_iteration": summaries[1]["seconds"] * 1e3 / max(summaries[1]["iterations"], 1),
           "stage2_iterations_to_within_1e-3_of_final_cost": it_1e3,
           # the headline of this block: stage 1 + stage 2 up to the iteration whose cost is
           # within 1e-3 of the final one; what follows is the harness solver wandering among the kinks
           "solve_ms_to_within_1e-3_of_final_cost": (summaries[0]["seconds"] + t_1e3) * 1e3,
           "registration_evaluation_ms": float(edt.item()) * 1e3,
           "position_rmse_m_odometry": rmse(poses0), "position_rmse_m_after_stage1": rmse(x1),
           "position_rmse_m_after": rmse(x),
           "position_rmse_m_aligned_odometry": rmse_aligned(poses0),
           "position_rmse_m_aligned_after_stage1": rmse_aligned(x1),
           "position_rmse_m_aligned_after": rmse_aligned(x),
           "rmse_note": "position_rmse_m_*: in the frame of the fixed first submap (the reference's gauge, "
                        "pose_graph_interface.cpp:30-32); *_aligned_*: after the best rigid alignment of the "
                        "whole estimate onto the truth (the registration cost is invariant to that transform "
                        "except through submap 0's own few constraints)",
           "stop_rule": "function_tolerance 1e-6 (Ceres default) in both stages, parameter_tolerance off",
           "parallelism": f"pair-sharded x{world}, submaps replicated, one all-reduce of "
                          f"{len(pairs) * capi.NORMAL_SIZE * 8} B (the per-constraint blocks) per evaluation",
           "setup_s": setup_s,
           "solver": "harness/lm.py (LM, banded Cholesky on the host; Ceres absent)"}
    if rank == 0 and not args.no_parity:
        from harness import parity_gate
        t_par = time.perf_counter()

        def layers_of(k):
            sm = capi.Submap.synth_city(ctx, k, vs, 16, bmin, dims, args.truncation, args.esdf_max, 10.0, true[k], args.seed)
            td, tw, ed, eo = sm.download_layers(16)
            bi = sm.block_index()
            sm.destroy()
            return bi, td, tw, ed, eo
        live_each = batch.count_live_each(poses0)
        chosen = parity_gate.choose(np.diff(batch.row_offsets()), live_each, n_total=8, n_partial=3, n_dead=1)
        e = parity_gate.check(capi, ctx, torch, "config 5 (the timed batch at the initial poses)", layers_of, submaps,
                              batch, pairs[mine], poses0, chosen, None, vs)
        e["seconds"] = time.perf_counter() - t_par
        out["parity"] = e
    for o in [batch] + cfs + submaps:
        o.destroy()
    return out

