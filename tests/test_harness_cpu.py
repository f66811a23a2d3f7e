"""CPU tests of the harness solver and of the N>1 path's host logic (gloo, world 2)."""
import os
import subprocess
import sys

import numpy as np

from harness import lm
from harness.backends import OracleBackend, assemble_fused
from oracle import pyoracle as orc
from oracle import synth
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_relative_pose_edge_jacobians_by_central_differences():
    """relative_pose_cost_function_inl.h:8-70 restated with analytic Jacobians."""
    rng = np.random.default_rng(0)
    poses = rng.normal(0, 1, (2, 4))
    e = lm.RelativePoseEdge(0, 1, rng.normal(0, 1, 3), 0.3, [1.0, 1.0, 2500.0, 2500.0])
    r, Ja, Jb = e.evaluate(poses)
    for blk, J in ((0, Ja), (1, Jb)):
        for k in range(4):
            p = poses.copy(); p[blk, k] += 1e-6
            m = poses.copy(); m[blk, k] -= 1e-6
            num = (e.evaluate(p)[0] - e.evaluate(m)[0]) / 2e-6
            np.testing.assert_allclose(J[:, k], num, rtol=1e-5, atol=1e-5)
    e2 = lm.RelativePoseEdge.from_poses(0, 1, poses[0], poses[1], [1, 1, 1, 1])
    assert np.abs(e2.evaluate(poses)[0]).max() < 1e-12
    assert -np.pi <= lm.normalize_angle(7.0) < np.pi


def _config1_problem():
    sm, _ = synth.config1_pair(asymmetric=True)
    layer = H.oracle_layer(sm)
    pts = H.oracle_points(sm)
    return OracleBackend([layer, layer], [pts, pts], [(0, 1)], 2, threads=1)


def test_known_answer_registration_recovers_duplicate_submap():
    """The reference's test-bench design (registration_test_bench.cpp:178-185,298-319):
    a duplicated submap perturbed on the yaml's grid must come back to the unperturbed
    pose.  CPU oracle backend, tight tolerances, < 1 mm / 0.01 deg."""
    backend = _config1_problem()
    grid = H.test_bench_grid(0.1)
    for pert in (grid[0], grid[13 + 27], grid[-1], np.array([0.15, -0.3, 0.15, 0.1])):
        poses0 = np.array([[0.0, 0, 0, 0], pert])
        prob = lm.Problem(backend, 2, [(0, 1)])
        x, s = lm.solve(prob, poses0, parameter_tolerance=1e-9, function_tolerance=1e-14,
                        max_iterations=60, max_seconds=60)
        assert np.abs(x[1, :3]).max() < 1e-3, (pert, x[1], s)
        assert abs(x[1, 3]) < np.deg2rad(0.01), (pert, x[1], s)
        assert s["final_cost"] <= 1e-6 * s["initial_cost"]


def test_assemble_fused_layout_matches_dense_normal_equations():
    rng = np.random.default_rng(1)
    pairs = [(0, 1), (1, 2), (0, 2)]
    normals, Jall = [], []
    for _ in pairs:
        J = rng.normal(0, 1, (50, 8)); r = rng.normal(0, 1, 50)
        Hm = J.T @ J
        normals.append(np.concatenate([[r @ r], J.T @ r, Hm[np.triu_indices(8)]]))
        Jall.append((J, r))
    cost, g, Hd = lm.unpack_fused(assemble_fused(normals, pairs, 3), 3, pairs)
    Jd = np.zeros((150, 12)); rd = np.zeros(150)
    for c, ((a, b), (J, r)) in enumerate(zip(pairs, Jall)):
        Jd[50 * c:50 * c + 50, 4 * a:4 * a + 4] = J[:, :4]
        Jd[50 * c:50 * c + 50, 4 * b:4 * b + 4] = J[:, 4:]
        rd[50 * c:50 * c + 50] = r
    np.testing.assert_allclose(cost, rd @ rd)
    np.testing.assert_allclose(g, Jd.T @ rd, atol=1e-12)
    np.testing.assert_allclose(Hd, Jd.T @ Jd, atol=1e-12)


def test_pair_sharding_allreduce_world2_gloo():
    """N > 1 host logic on CPU: two gloo ranks each assemble their LPT shard of the
    constraints (per-constraint normals from the CPU oracle standing in for the
    kernel), all-reduce the fused buffer, and must reproduce the unsharded one."""
    script = os.path.join(ROOT, "tests", "_gloo_shard_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                          "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port",
                          "29517", script], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "SHARD_ALLREDUCE_OK" in out.stdout


def test_vectorised_problem_matches_reference_assembly():
    """Problem.evaluate (scatter indices, batched edges) == unpack_fused + per-edge loop."""
    rng = np.random.default_rng(7)
    n, pairs = 5, [(0, 1), (1, 0), (1, 2), (3, 4), (0, 4), (2, 3)]
    normals = []
    for _ in pairs:
        J = rng.normal(0, 1, (30, 8)); r = rng.normal(0, 1, 30)
        normals.append(np.concatenate([[r @ r], J.T @ r, (J.T @ J)[np.triu_indices(8)]]))
    buf = assemble_fused(normals, pairs, n)
    poses = rng.normal(0, 2, (n, 4))
    edges = [lm.RelativePoseEdge(k, k + 1, rng.normal(0, 1, 3), rng.normal(0, 0.5),
                                 [1.0, 1.0, 2500.0, 2500.0]) for k in range(n - 1)]
    prob = lm.Problem(lambda p: buf, n, pairs, edges)
    cost, g, Hm = prob.evaluate(poses)
    c0, g0, H0 = lm.unpack_fused(buf, n, pairs)
    for e in edges:
        r, Ja, Jb = e.evaluate(poses)
        c0 += r @ r
        ia, ib = slice(4 * e.a, 4 * e.a + 4), slice(4 * e.b, 4 * e.b + 4)
        g0[ia] += Ja.T @ r; g0[ib] += Jb.T @ r
        H0[ia, ia] += Ja.T @ Ja; H0[ib, ib] += Jb.T @ Jb
        H0[ia, ib] += Ja.T @ Jb; H0[ib, ia] += Jb.T @ Ja
    np.testing.assert_allclose(cost, 0.5 * c0, rtol=1e-12)
    np.testing.assert_allclose(g, g0, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(Hm, H0, rtol=1e-10, atol=1e-10)
    assert np.allclose(Hm, Hm.T)


def test_reduced_banded_form_matches_dense_normal_equations():
    """evaluate_reduced (RCM order, banded storage) == evaluate (dense) on the free block."""
    from scipy.linalg import solveh_banded
    rng = np.random.default_rng(9)
    n, pairs = 6, [(0, 1), (1, 0), (1, 2), (3, 4), (0, 4), (2, 3), (4, 5), (5, 2)]
    normals = []
    for _ in pairs:
        J = rng.normal(0, 1, (40, 8)); r = rng.normal(0, 1, 40)
        normals.append(np.concatenate([[r @ r], J.T @ r, (J.T @ J)[np.triu_indices(8)]]))
    buf = assemble_fused(normals, pairs, n)
    poses = rng.normal(0, 2, (n, 4))
    edges = [lm.RelativePoseEdge(k, k + 1, rng.normal(0, 1, 3), rng.normal(0, 0.5),
                                 [1.0, 1.0, 2500.0, 2500.0]) for k in range(n - 1)]
    for const in ((0,), (2,), (0, 5)):
        prob = lm.Problem(lambda p: buf, n, pairs, edges, constant_nodes=const)
        cost, g, Hd = prob.evaluate(poses)
        cost2, gf, (band, V) = prob.evaluate_reduced(poses)
        f = prob._perm_full
        assert sorted(f) == sorted(prob.free)
        np.testing.assert_allclose(cost2, cost, rtol=1e-12)
        np.testing.assert_allclose(gf, g[f], rtol=1e-12, atol=1e-12)
        Hf = Hd[np.ix_(f, f)]
        u = prob._u
        dense_from_band = np.zeros_like(Hf)
        for i in range(len(f)):
            for j in range(i, min(len(f), i + u + 1)):
                dense_from_band[i, j] = dense_from_band[j, i] = band[u + i - j, j]
        np.testing.assert_allclose(dense_from_band, Hf, rtol=1e-10, atol=1e-9)
        xv = rng.normal(0, 1, len(f))
        np.testing.assert_allclose(prob.reduced_matvec(V, xv), Hf @ xv, rtol=1e-10, atol=1e-9)
        A = band.copy(); A[u] += 10.0
        np.testing.assert_allclose(solveh_banded(A, gf, lower=False),
                                   np.linalg.solve(Hf + 10 * np.eye(len(f)), gf), rtol=1e-8, atol=1e-10)


def test_two_stage_optimise_with_loop_closure_pulls_in_accumulated_drift():
    """BASELINE config 5 in miniature (pose_graph_interface.cpp:177-198): a loop of submaps
    with odometry drift far outside the registration basin; the loop-closure edge alone
    (stage 1, registration excluded) brings the chain back, stage 2 refines with
    registration (CPU oracle backend)."""
    sdf = synth.union_sdf(synth.sphere_ground_sdf((1.6, 1.6, 1.2), 1.0, 0.35),
                          synth.sphere_sdf((0.6, 2.4, 0.8), 0.5))
    n = 6
    ang = np.linspace(0, 2 * np.pi, n, endpoint=False)
    true = np.stack([0.5 * np.cos(ang) - 0.5, 0.5 * np.sin(ang), 0.02 * np.arange(n), 0.1 * np.sin(ang)], 1)
    true[0] = 0
    layers, pts = [], []
    for p in true:
        sm = synth.make_submap(sdf, 0.1, 16, (0, 0, 0), (2, 2, 2), 0.3, p, 1.0, drop_empty_blocks=True)
        layers.append(H.oracle_layer(sm))
        pts.append(H.oracle_points(sm))
    pairs = [(k, (k + 1) % n) for k in range(n)]
    # odometry: true relative motion plus a bias that accumulates around the loop
    info_odo, info_lc = [1.0, 1.0, 2500.0, 2500.0], [100.0, 100.0, 2500.0, 2500.0]
    poses0 = true.copy()
    for k in range(1, n):
        poses0[k] = poses0[k - 1] + (true[k] - true[k - 1]) + np.array([0.12, -0.08, 0.0, 0.03])
    edges = [lm.RelativePoseEdge.from_poses(k, k + 1, poses0[k], poses0[k + 1], info_odo) for k in range(n - 1)]
    edges.append(lm.RelativePoseEdge.from_poses(n - 1, 0, true[n - 1], true[0], info_lc))   # the loop closure
    backend = OracleBackend(layers, pts, pairs, n)
    kw = dict(parameter_tolerance=1e-8, function_tolerance=1e-10, max_iterations=60, max_seconds=120)
    drift0 = np.abs(poses0[:, :2] - true[:, :2]).max()
    assert drift0 > 0.4                                   # well outside the 0.3 m band
    x2, summ = lm.optimize_two_stage(backend, n, pairs, edges, poses0, True, **kw)
    assert len(summ) == 2
    e2 = np.abs(x2[:, :3] - true[:, :3]).max()
    assert e2 < 0.05 and e2 < 0.2 * drift0, (e2, drift0)


def test_solve_survives_single_thread_blas_environment():
    """torch.distributed.run exports OMP_NUM_THREADS=1; an OpenBLAS that starts with one thread
    crashes if a thread-pool limit later RAISES it (seen at N=2 on the GPU box).  lm.solve must
    only ever lower the pools."""
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import torch, numpy as np\n"
        "from harness import lm\n"
        "n = 60; rng = np.random.default_rng(0)\n"
        "true = np.c_[np.arange(n) * 2.0, rng.normal(0, 0.1, (n, 3))]\n"
        "edges = [lm.RelativePoseEdge.from_poses(k, k + 1, true[k], true[k + 1], [1, 1, 2500, 2500])\n"
        "         for k in range(n - 1)]\n"
        "prob = lm.Problem(lm.zero_registration_backend(n, 0), n, np.zeros((0, 2), np.int64), edges)\n"
        "x, s = lm.solve(prob, true + rng.normal(0, 0.05, true.shape), parameter_tolerance=1e-10)\n"
        "assert s['final_cost'] < 1e-12, s\n"
        "print('SOLVED')\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "SOLVED" in r.stdout, (r.returncode, r.stderr[-2000:])


def test_parity_gate_picks_culled_and_live_constraints():
    """bench.py's same-run parity gate samples deterministically: a fully culled constraint when there is
    one, partly culled ones spread over the culled fraction, the rest the heaviest live ones"""
    from harness import parity_gate
    n_rows = np.array([1000, 2000, 3000, 4000, 5000, 6000, 7000, 8000, 9000, 10000])
    live = np.array([0, 2000, 300, 4000, 2500, 6000, 0, 7900, 4500, 10000])
    got = parity_gate.choose(n_rows, live, n_total=6, n_partial=3, n_dead=1)
    assert len(got) == len(set(got)) == 6
    assert got[0] == 0                                            # the first fully culled one
    partial = [c for c in got if 0 < live[c] < 0.9 * n_rows[c]]
    assert len(partial) >= 3 and 2 in partial and 4 in partial     # least and most live of the partly culled
    full = [c for c in got if live[c] >= 0.9 * n_rows[c]]
    assert full and full[0] == 9                                   # heaviest fully live first
    assert parity_gate.choose(n_rows, live, 6, 3, 1) == got        # deterministic
    # fewer constraints than asked for: all of them, once
    assert sorted(parity_gate.choose(n_rows[:3], live[:3], 8, 3, 1)) == [0, 1, 2]
    m = parity_gate.merge([dict(values_checked=10, constraints_checked=2, max_rel=0.0, exact=True, fused_blocks_max_rel=1e-7,
                                **{"fused_blocks_within_1e-6": True}, checker="x", rule="y"),
                           dict(values_checked=5, constraints_checked=1, max_rel=1e-3, exact=False, fused_blocks_max_rel=2e-6,
                                **{"fused_blocks_within_1e-6": False}, checker="x", rule="y"), None])
    assert m["checked"] == 15 and not m["exact"] and m["max_rel"] == 1e-3 and not m["fused_blocks_within_1e-6"]
    assert parity_gate.merge([None]) is None


def test_profiled_launch_counts_come_from_the_committed_trace():
    """bench.py's `tsdf.*.{reproducible_mode,merged_integrator}.roofline.launches_per_scan_from_profiles`: parsed from
    profiles/r06_tsdf_launches.txt (rocprofv3 kernel trace of one scan of each sort-based path, tsdf_launches.sh) -- and
    only while that trace describes the sources in the tree: the script records a hash of the TSDF sources,
    a mismatch drops the counts instead of dividing a fresh time by a stale number"""
    import os
    from harness import bench_tsdf
    got, note = bench_tsdf.profiled_launches()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "profiles", "r06_tsdf_launches.txt")
    recorded = [l.split()[-1] for l in open(path) if l.startswith("sources sha256:")] if os.path.exists(path) else []
    if recorded and recorded[0] == bench_tsdf.tsdf_sources_sha():
        assert set(got) == {("fast", "lidar"), ("fast", "rgbd"), ("merged", "lidar"), ("merged", "rgbd")}, note
        # the paths are launch bound: these are the numbers rounds 4 and 5 worked on (78 / 93 / 30 / 49 when round 4 began)
        assert got[("fast", "lidar")] <= 44 and got[("fast", "rgbd")] <= 55
        assert got[("merged", "lidar")] <= 26 and got[("merged", "rgbd")] <= 38
        assert all(v > 5 for v in got.values())
    else:
        assert got == {} and ("other sources" in note or "no committed trace" in note), note


def test_box_state_degrades_gracefully_without_a_gpu_and_the_line_carries_the_ceilings():
    """harness/box_state.py (which box was it?) must never cost the line: without amdgpu in sysfs /
    without rocm-smi it reports `available: False`; and harness/bench_line.py passes the same-run ceilings and the box's
    clocks through to the compact line"""
    from harness import bench_line, box_state
    snap = box_state.snapshot()
    assert set(snap) == {"sysfs", "rocm_smi", "firmware"} and "available" in snap["firmware"] and "available" in snap["sysfs"] and "available" in snap["rocm_smi"]
    with box_state.Sampler(period_s=0.001) as s:
        pass
    assert "available" in s.summary()
    full = {"metric": "m", "value": 1.0, "unit": "u", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": 1.0,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "w", "output_placement": "row arrays: the fastest of 4 allocations of each kind"},
            "tsdf": {"lidar_64x1024": {"ms_per_scan": 0.021, "ms_per_scan_fresh_integrator": 0.044, "integrator_age_scans": 100,
                                       "reproducible_mode": {"ms_per_scan": 0.25, "ms_per_scan_fresh_integrator": 0.31},
                                       "cpu_baseline": {"Mpoints_per_s": 70.0, "Mpoints_per_s_fresh_integrator": 54.0}}},
            "roofline": {"bound": "hbm", "achieved": 6750.0, "placement_ms_sets": [5.59, 4.36, 4.36, 4.43], "peak": 8000.0, "unit": "GB/s", "frac": 0.84, "traffic": 3.0e10,
                         "traffic_source": "profiles/hbm_traffic.json", "hbm_frac": 0.79, "copy_ceiling_GBs": 6315.0,
                         "fill_ceiling_GBs": 6778.0, "kernel_shaped_ceiling_GBs": 6273.0, "frac_of_copy_ceiling": 1.07,
                         "traffic_frac_of_copy_ceiling": 1.01, "frac_of_kernel_shaped_ceiling": 1.08, "kernel_ms": 4.7},
            "box": {"before": {"sysfs": {"power_cap_W": 1400.0, "compute_partition": "SPX", "memory_partition": "NPS1",
                                         "pci_bus_id": "0000:0d:00.0", "vbios": "113-M355-01-1K1-030A"}},
                    "during_timed_region": {"sclk_MHz": {"median": 2390.0}, "mclk_MHz": {"median": 2000.0},
                                            "power_in_W": {"median": 1152.0}}}}
    out, line = bench_line.compact(full, "bench_detail.json")
    assert out["roofline"]["copy_ceiling_GBs"] == 6315.0 and out["roofline"]["traffic_frac_of_copy_ceiling"] == 1.01
    assert out["roofline"]["traffic_source"] == "profiles/hbm_traffic.json"
    assert out["box"] == {"sclk_MHz_during": 2390.0, "mclk_MHz_during": 2000.0, "power_W_during": 1152.0, "power_cap_W": 1400.0,
                          "pci_bus_id": "0000:0d:00.0", "vbios": "113-M355-01-1K1-030A",
                          "compute_partition": "SPX", "memory_partition": "NPS1"}
    # the two findings of round 5's end travel in the line: what the candidate placements of the row arrays cost, and
    # the session-old integrator's TSDF figures next to the fresh one's
    assert out["roofline"]["placement_ms_sets"] == [5.59, 4.36, 4.36, 4.43] and "fastest of 4" in out["config"]["output_placement"]
    t = out["tsdf"]["lidar"]
    assert (t["ms_per_scan"], t["ms_per_scan_fresh_integrator"], t["integrator_age_scans"]) == (0.021, 0.044, 100)
    assert (t["reproducible_ms_per_scan"], t["reproducible_ms_per_scan_fresh_integrator"]) == (0.25, 0.31)
    assert (t["cpu_Mpoints_per_s_1_core"], t["cpu_Mpoints_per_s_1_core_fresh_integrator"]) == (70.0, 54.0)
    assert len(line) < bench_line.LINE_LIMIT
