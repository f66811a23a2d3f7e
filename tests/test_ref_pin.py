"""Pins the REG oracle (oracle/reg_oracle.c) to the REFERENCE'S OWN SOURCE.

oracle/_ref/libref_reg.so is /root/reference's registration_cost_function.cpp (plus its
weighted_sampler / registration_point headers) compiled against the stand-in headers of
oracle/ref_shims (Eigen, glog, minkindr, voxblox, Ceres, ROS are absent from this image).
tests/golden/ref_reg_config1.npz holds its outputs (tests/golden/make_ref_golden.py).

  * golden test  -- runs everywhere: the oracle must reproduce the stored outputs bit for bit
  * live tests   -- run where the library exists: a wider randomised differential

What remains recalled (not pinned) is what the shims restate: voxblox's interpolator/grid and
minkindr's transformation."""
import hashlib
import os

import numpy as np
import pytest

from oracle import pyoracle as orc
from oracle import ref_reg, synth
from tests import helpers as H

F = np.float32


def _digest(a):
    return hashlib.sha256(np.ascontiguousarray(np.asarray(a) + 0.0).tobytes()).hexdigest()   # -0 == +0


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "ref_reg_config1.npz"))


@pytest.fixture(scope="module")
def scene():
    ref, read = synth.config1_pair(seed=0, asymmetric=True)
    pts = {e: orc.find_relevant_voxels(ref.voxel_size, ref.vps, ref.block_index, ref.tsdf_distance,
                                       ref.tsdf_weight, ref.esdf_distance if e else None, 1.0, 0.3)
           for e in (True, False)}
    return ref, read, pts


def sequential_cumsum(w):
    """WeightedSampler::addItem accumulates one item at a time in double."""
    out, acc = np.zeros(len(w)), 0.0
    for i, v in enumerate(np.asarray(w, np.float64)):
        acc = v if i == 0 else acc + v
        out[i] = acc
    return out


GOLDEN_CASES = [("esdf_all", True, 0.0, None), ("tsdf_nocorr", False, 0.7, None),
                ("esdf_sampled", True, 0.0, 0.05)]


def test_golden_inputs_are_the_ones_the_fixture_was_made_from(golden, scene):
    ref, read, pts = scene
    for name, arr in (("ref_tsdf", ref.tsdf_distance), ("ref_esdf", ref.esdf_distance),
                      ("read_tsdf", read.tsdf_distance), ("read_esdf", read.esdf_distance),
                      ("points_esdf_xyz", pts[True][0]), ("points_tsdf_d", pts[False][1])):
        assert str(golden["input_sha_" + name]) == _digest(arr), name


@pytest.mark.parametrize("case,use_esdf,no_corr,ratio", GOLDEN_CASES)
def test_oracle_reproduces_the_reference_source_bit_for_bit(golden, scene, case, use_esdf, no_corr, ratio):
    ref, read, pts = scene
    xyz, d, w = pts[use_esdf]
    layer = H.oracle_layer(read, use_esdf)
    n = int(golden[f"{case}_num_residuals"])
    base, stride = golden["base_pose"], int(golden["stride"])
    eng, cum = orc.Mt19937(5489), (sequential_cumsum(w) if ratio else None)
    if ratio:
        assert n == int(F(ratio) * F(len(w)))                     # .cpp:49-50 (float product, truncated)
    else:
        assert n == len(w)
    for k, pert in enumerate(golden["perturbations"]):
        idx = np.array([eng.weighted_draw(cum) for _ in range(n)], np.int64) if ratio else None
        ok, r, j0, j1 = orc.reg_evaluate(layer, xyz, d, w, base, base + pert,
                                         no_correspondence_cost=no_corr, sample_idx=idx)
        assert ok
        key = f"{case}_{k}"
        assert np.array_equal(r[::stride], golden[key + "_r"])
        assert np.array_equal(j0[::stride], golden[key + "_jref"])
        assert np.array_equal(j1[::stride], golden[key + "_jread"])
        assert [_digest(r), _digest(j0), _digest(j1)] == list(golden[key + "_sha"]), key
        assert int((np.abs(j0).sum(1) > 0).sum()) == int(golden[key + "_corr"])


needs_ref = pytest.mark.skipif(not ref_reg.available(),
                               reason="oracle/_ref/libref_reg.so not built (needs /root/reference)")


def _ref_pair(scene, use_esdf, point_type=ref_reg.POINTS_VOXELS, points=None):
    ref, read, pts = scene
    xyz, d, w = points if points is not None else pts[use_esdf]
    R = ref_reg.Submap(0, ref.pose, ref.voxel_size, ref.vps, ref.block_index, ref.tsdf_distance,
                       ref.tsdf_weight, ref.esdf_distance, ref.esdf_observed)
    R.set_points(point_type, xyz, d, w)
    E = ref_reg.Submap(1, read.pose, read.voxel_size, read.vps, read.block_index, read.tsdf_distance,
                       read.tsdf_weight, read.esdf_distance, read.esdf_observed)
    return R, E, (xyz, d, w)


@needs_ref
def test_live_reference_over_the_test_bench_grid(scene):
    """config/registration_test_bench.yaml:9-13 grid, both distance modes: bit-exact."""
    ref, read, _ = scene
    for use_esdf in (True, False):
        R, E, (xyz, d, w) = _ref_pair(scene, use_esdf)
        cf = ref_reg.RegistrationCostFunction(R, E, use_esdf_distance=use_esdf)
        layer = H.oracle_layer(read, use_esdf)
        base = np.array([1.3, -0.7, 0.2, 0.4])
        for pert in H.test_bench_grid(ref.voxel_size)[::2]:
            ok1, r1, a1, b1 = cf.Evaluate(base, base + pert)
            ok2, r2, a2, b2 = orc.reg_evaluate(layer, xyz, d, w, base, base + pert)
            assert ok1 and ok2
            assert np.array_equal(r1, r2) and np.array_equal(a1, a2) and np.array_equal(b1, b2)


@needs_ref
def test_live_reference_randomised_configs(scene):
    """random poses (large translations and yaws included), no-correspondence costs, null Jacobian
    blocks, random per-point weights, isosurface-type point sets"""
    ref, read, pts = scene
    rng = np.random.default_rng(5)
    for trial in range(12):
        use_esdf = bool(trial % 2)
        xyz, d, w = pts[use_esdf]
        keep = rng.random(len(w)) < 0.3
        pw = (w[keep] * rng.uniform(0.05, 2.0, keep.sum())).astype(F)
        ptype = ref_reg.POINTS_ISOSURFACE if trial % 3 == 0 else ref_reg.POINTS_VOXELS
        # off-grid positions, as isosurface vertices are
        pxyz = (xyz[keep] + rng.uniform(-0.04, 0.04, (keep.sum(), 3))).astype(F)
        R, E, _ = _ref_pair(scene, use_esdf, ptype, (pxyz, d[keep], pw))
        no_corr = float(rng.choice([0.0, 0.3, 2.0]))
        cf = ref_reg.RegistrationCostFunction(R, E, ptype, no_correspondence_cost=no_corr,
                                              use_esdf_distance=use_esdf)
        layer = H.oracle_layer(read, use_esdf)
        a = np.r_[rng.uniform(-50, 50, 2), rng.uniform(-2, 2), rng.uniform(-3.1, 3.1)]
        b = a + np.r_[rng.normal(0, 0.3, 3), rng.normal(0, 0.15)]
        want_ref, want_read = bool(trial % 4 != 1), bool(trial % 4 != 2)
        want_jac = trial != 7
        ok1, r1, a1, b1 = cf.Evaluate(a, b, want_jac, want_ref, want_read)
        ok2, r2, a2, b2 = orc.reg_evaluate(layer, pxyz, d[keep], pw, a, b, want_jac, want_ref, want_read,
                                           no_correspondence_cost=no_corr)
        assert ok1 and ok2
        assert np.array_equal(r1, r2), trial
        for x, y in ((a1, a2), (b1, b2)):
            assert (x is None) == (y is None)
            if x is not None:
                assert np.array_equal(x, y), trial


@needs_ref
def test_live_reference_sampler_stream_and_zero_weight(scene):
    """sampling_ratio != -1: the same draws in the same order, call after call; and Evaluate
    returns false when the summed weight is zero (.cpp:273)"""
    ref, read, pts = scene
    xyz, d, w = pts[True]
    rng = np.random.default_rng(9)
    w2 = (w * rng.uniform(0.1, 1.0, len(w))).astype(F)
    R, E, _ = _ref_pair(scene, True, points=(xyz, d, w2))
    cf = ref_reg.RegistrationCostFunction(R, E, sampling_ratio=0.1)
    n = cf.num_residuals()
    assert n == int(F(0.1) * F(len(w2)))
    layer, eng, cum = H.oracle_layer(read, True), orc.Mt19937(5489), sequential_cumsum(w2)
    a, b = np.array([0.0, 0.1, 0.0, 0.02]), np.array([0.05, 0.0, 0.02, -0.03])
    for call in range(3):
        idx = np.array([eng.weighted_draw(cum) for _ in range(n)], np.int64)
        ok1, r1, a1, b1 = cf.Evaluate(a, b)
        ok2, r2, a2, b2 = orc.reg_evaluate(layer, xyz, d, w2, a, b, sample_idx=idx)
        assert ok1 and ok2
        assert np.array_equal(r1, r2) and np.array_equal(a1, a2) and np.array_equal(b1, b2), call
    R0, E0, _ = _ref_pair(scene, True, points=(xyz[:100], d[:100], np.zeros(100, F)))
    cf0 = ref_reg.RegistrationCostFunction(R0, E0)
    ok1 = cf0.Evaluate(a, b)[0]
    ok2 = orc.reg_evaluate(layer, xyz[:100], d[:100], np.zeros(100, F), a, b)[0]
    assert ok1 is False and ok2 is False


@needs_ref
def test_harness_relative_pose_edge_is_the_reference_functor():
    """harness/lm.py's odometry / loop-closure edge against the reference's RelativePoseCostFunction
    (relative_pose_cost_function_inl.h + normalize_angle.h, T = double, built through its own
    Create()): with the registration cost pinned too, the stand-in solver minimises voxgraph's
    objective itself; only the solver differs from Ceres."""
    from harness import lm
    rng = np.random.default_rng(4)
    worst = 0.0
    for trial in range(200):
        observed = np.r_[rng.uniform(-8, 8, 3), rng.uniform(-3.0, 3.0)]
        info_diag = rng.choice([1.0, 2500.0, 0.25, 100.0], 4)                 # voxgraph_mapper.yaml:41-47 style
        a = np.r_[rng.uniform(-50, 50, 3), rng.uniform(-3.1, 3.1)]
        b = np.r_[a[:3] + rng.uniform(-9, 9, 3), rng.uniform(-3.1, 3.1)]    # CHECK_NEAR(yaw, 0, pi) holds
        r_ref, stored = ref_reg.relative_pose_residual(observed, np.diag(np.sqrt(info_diag)), a, b)
        edge = lm.RelativePoseEdge(0, 1, stored[:3], stored[3], info_diag)
        r, _, _ = edge.evaluate(np.array([a, b]))
        worst = max(worst, float(np.abs(r - r_ref).max() / max(1.0, np.abs(r_ref).max())))
    assert worst < 1e-12, worst
