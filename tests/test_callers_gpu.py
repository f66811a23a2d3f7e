"""The reference's OWN callers on top of either cost function: pose_graph.cpp,
constraint_collection.cpp, constraint.cpp, relative_pose_constraint.cpp, absolute_pose_constraint.cpp (with
relative_pose_cost_function_inl.h differentiated by Jets), registration_constraint.cpp, node.cpp, node_collection.cpp,
pose_4d.cpp and submap_registration_helper.cpp are compiled from /root/reference where they lie (oracle/Makefile: ref)
-- once as they are, once with the ONE edit INTEGRATION.md section 3 shows applied by sed at build time
(`new RegistrationCostFunction(` -> `voxgraph_amd::MakeGpuRegistrationCostFunction(`, gpu_submap_registry.h) -- and both
binaries run PoseGraph::optimize() on a four-submap graph as voxgraph builds it (kVoxels and mirrored kIsosurfacePoints
registration constraints, odometry edges with the shipped information matrix, a semi-definite height measurement),
SubmapRegistrationHelper::testRegistration(), the alignment problem of map_evaluation.cpp:116-161 (restated: that
file is a ROS node), PoseGraph::getEdgeCovarianceMap() (ceres::Covariance over the cost functions' Jacobians at the final
poses: pose_graph.cpp:117-163, served to loop_closure_edge_server.cpp:46) and the two-stage optimisation after a loop
closure (pose_graph_interface.cpp:182-191).  No hand-written stand-in for reference code is left in oracle/ref_driver/callers_check.cpp.  The
solver behind ceres::Solve is the stand-in of tests/stubs/ceres (the real Ceres is absent from this image).

Bar: the same final poses within 1 mm / 0.01 deg (north_star).  The binaries travel to the GPU box with the snapshot."""
import math
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "callers_check_reference")
GPU = os.path.join(ROOT, "oracle", "_ref", "callers_check_gpu")
TRUTH = {10: (0.0, 0.0, 0.0, 0.0), 11: (1.3, 0.2, 0.02, 0.06), 12: (2.5, -0.1, 0.0, -0.05), 13: (3.4, 0.15, -0.03, 0.04)}


def _run(binary):
    r = subprocess.run([binary], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    poses = {(int(m.group(1)), int(m.group(2))): tuple(float(x) for x in m.group(3).split())
             for m in re.finditer(r"POSE point_type=(\d) submap=(\d+) (.*)", r.stdout)}
    solves = {int(m.group(1)): (int(m.group(2)), float(m.group(3)), float(m.group(4)))
              for m in re.finditer(r"SOLVE point_type=(\d) iterations=(\d+) initial_cost=(\S+) final_cost=(\S+)", r.stdout)}
    edges = {int(m.group(1)): float(m.group(2)) for m in re.finditer(r"EDGES point_type=(\d) sum_sq_residuals=(\S+)", r.stdout)}
    m = re.search(r"HELPER usable=(\d) iterations=(\d+) final_cost=(\S+) pose (.*)", r.stdout)
    helper = (int(m.group(1)), int(m.group(2)), float(m.group(3)), tuple(float(x) for x in m.group(4).split()))
    m = re.search(r"ALIGN iterations=(\d+) final_cost=(\S+) pose (.*)", r.stdout)
    align = (int(m.group(1)), float(m.group(2)), tuple(float(x) for x in m.group(3).split()))
    assert len(poses) == 8 and len(solves) == 2 and len(edges) == 2
    return poses, solves, edges, helper, align


def _run_more(binary):
    """the later additions to callers_check: PoseGraph::getEdgeCovarianceMap after optimize() (pose_graph.cpp:117-163) and
    the two-stage optimisation after a loop closure (pose_graph_interface.cpp:182-191: optimize(true), then optimize())"""
    r = subprocess.run([binary], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    cov_ok = {int(m.group(1)): (int(m.group(2)), int(m.group(3))) for m in re.finditer(r"COVARIANCE point_type=(\d) ok=(\d) pairs=(\d+)", r.stdout)}
    cov = {(int(m.group(1)), int(m.group(2)), int(m.group(3))): tuple(float(x) for x in m.group(4).split())
           for m in re.finditer(r"COV point_type=(\d) pair=(\d+),(\d+) (.*)", r.stdout)}
    m = re.search(r"TWOSTAGE iterations=(\d+),(\d+) final_cost=(\S+),(\S+)", r.stdout)
    two = (int(m.group(1)), int(m.group(2)), float(m.group(3)), float(m.group(4)))
    poses2 = {int(m.group(1)): tuple(float(x) for x in m.group(2).split()) for m in re.finditer(r"POSE2 submap=(\d+) (.*)", r.stdout)}
    assert cov_ok == {0: (1, 5), 1: (1, 5)} and len(cov) == 10 and all(len(v) == 16 for v in cov.values()) and len(poses2) == 4
    return cov, two, poses2


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/callers_check_reference not built (needs /root/reference)")
def test_reference_callers_run_on_the_reference_cost_function():
    """(no GPU needed) the reference's PoseGraph::optimize(), compiled from its sources, pulls the drifted graph back"""
    poses, solves, edges, helper, align = _run(REF)
    assert max(abs(a - b) for a, b in zip(align[2][:3], TRUTH[11][:3])) < 0.012 and abs(align[2][3] - TRUTH[11][3]) < 0.006, align
    for pt in (0, 1):
        its, c0, c1 = solves[pt]
        assert c1 < 0.1 * c0 and its >= 1
        for sid, want in TRUTH.items():
            got = poses[(pt, sid)]
            assert max(abs(a - b) for a, b in zip(got[:3], want[:3])) < 0.012 and abs(got[3] - want[3]) < 0.006, (pt, sid, got)
        assert edges[pt] == pytest.approx(2.0 * c1, rel=1e-9)      # the edges' squared residuals ARE the cost
    assert helper[0] == 1
    cov, two, poses2 = _run_more(REF)
    for (pt, a, b), v in cov.items():
        if a == 10:   # the first submap is constant (pose_graph_interface.cpp:30-32): no covariance with it
            assert all(x == 0.0 for x in v)
        else:         # the cross-covariance of two free poses: finite, not all zero
            assert all(math.isfinite(x) for x in v) and any(x != 0.0 for x in v)
    assert two[0] >= 1 and two[1] >= 1
    for sid, want in TRUTH.items():
        got = poses2[sid]
        assert max(abs(a - b) for a, b in zip(got[:3], want[:3])) < 0.012 and abs(got[3] - want[3]) < 0.006, (sid, got)


@pytest.mark.gpu
@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(GPU)),
                    reason="oracle/_ref/callers_check_* not built (needs /root/reference)")
def test_pose_graph_optimize_gives_the_same_poses_on_the_gpu_cost_function():
    ref_poses, ref_solves, ref_edges, ref_helper, ref_align = _run(REF)
    gpu_poses, gpu_solves, gpu_edges, gpu_helper, gpu_align = _run(GPU)
    # GpuSubmapRegistry: the cached upload is reused for the same object and stamp, redone when the object is
    # finished again (stamp change), and an entry whose owner died without release() is swept
    out = subprocess.run([GPU], capture_output=True, text=True, timeout=600).stdout
    m = re.search(r"REGISTRY cached=(\d) stamp_change_reuploaded=(\d) stale_replaced=(\d+) size_before=(\d+) size_with_temp=(\d+) "
                  r"size_after_sweep=(\d+)", out)
    assert m, out[-1500:]
    cached, reup, stale, before, with_temp, after = (int(x) for x in m.groups())
    assert cached == 1 and reup == 1 and stale == 1 and with_temp == before + 1 and after == before
    worst_m = worst_rad = 0.0
    for key, want in ref_poses.items():
        got = gpu_poses[key]
        worst_m = max(worst_m, max(abs(a - b) for a, b in zip(got[:3], want[:3])))
        worst_rad = max(worst_rad, abs(got[3] - want[3]))
    worst_m = max(worst_m, max(abs(a - b) for a, b in zip(gpu_helper[3][:3], ref_helper[3][:3])))
    worst_rad = max(worst_rad, abs(gpu_helper[3][3] - ref_helper[3][3]))
    worst_m = max(worst_m, max(abs(a - b) for a, b in zip(gpu_align[2][:3], ref_align[2][:3])))
    worst_rad = max(worst_rad, abs(gpu_align[2][3] - ref_align[2][3]))
    assert gpu_align[0] == ref_align[0] and gpu_align[1] == pytest.approx(ref_align[1], rel=1e-4)
    print(f"worst pose difference: {worst_m * 1e3:.6f} mm, {math.degrees(worst_rad):.6f} deg")
    assert worst_m < 1e-3 and worst_rad < math.radians(0.01)
    for pt in (0, 1):
        assert gpu_solves[pt][0] == ref_solves[pt][0]                                  # the same iterations
        assert gpu_solves[pt][1] == pytest.approx(ref_solves[pt][1], rel=1e-6)         # initial cost
        assert gpu_solves[pt][2] == pytest.approx(ref_solves[pt][2], rel=1e-4)
        assert gpu_edges[pt] == pytest.approx(ref_edges[pt], rel=1e-4)
    assert gpu_helper[0] == ref_helper[0] == 1 and gpu_helper[1] == ref_helper[1]
    # the covariance blocks Ceres extracts from the cost functions' Jacobians at the final poses, and the two-stage solve
    ref_cov, ref_two, ref_poses2 = _run_more(REF)
    gpu_cov, gpu_two, gpu_poses2 = _run_more(GPU)
    worst_cov = 0.0
    for key, want in ref_cov.items():
        got = gpu_cov[key]
        scale = max(abs(x) for x in want)
        if scale == 0.0:
            assert all(x == 0.0 for x in got), key
            continue
        worst_cov = max(worst_cov, max(abs(a - b) for a, b in zip(got, want)) / scale)
    print(f"worst covariance entry difference, relative to the block's largest entry: {worst_cov:.3e}")
    assert worst_cov < 1e-4
    assert gpu_two[:2] == ref_two[:2] and gpu_two[2] == pytest.approx(ref_two[2], rel=1e-4) and gpu_two[3] == pytest.approx(ref_two[3], rel=1e-4)
    for sid, want in ref_poses2.items():
        got = gpu_poses2[sid]
        assert max(abs(a - b) for a, b in zip(got[:3], want[:3])) < 1e-3 and abs(got[3] - want[3]) < math.radians(0.01), (sid, got, want)
