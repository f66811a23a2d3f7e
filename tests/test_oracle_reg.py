"""Pins the CPU oracle (oracle/reg_oracle.c) before anything is compared to it.

The reference has no tests or fixtures for this path, so the
oracle is pinned against: the reference's own sympy derivation (golden JSON),
libstdc++'s mt19937/uniform_real_distribution (the library the reference
samples with), closed forms on a planar SDF and central differences.
"""
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

from oracle import pyoracle as orc
from oracle import synth

F = np.float32


# ---------------------------------------------------------------- golden -----
def _golden(golden_dir):
    with open(os.path.join(golden_dir, "jacobians_xyz_yaw.json")) as fh:
        return json.load(fh)["cases"]


def test_pose_jacobian_matrices_match_reference_sympy(golden_dir):
    """registration_cost_function.cpp:214-227 == scripts/jacobians_xyz_yaw.py."""
    for c in _golden(golden_dir):
        mo, me = orc.pose_jacobian_matrices(c["point"][0], c["point"][1], c["ref_pose"],
                                            c["read_pose"])
        np.testing.assert_allclose(mo, np.array(c["M_ref"]), rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(me, np.array(c["M_read"]), rtol=2e-5, atol=5e-5)


def test_relative_transform_matches_reference_sympy(golden_dir):
    """exp(reading)^-1 * exp(reference) applied to a point == the script's T_eo * r."""
    for c in _golden(golden_dir):
        q, t = orc.relative_transform(c["ref_pose"], c["read_pose"])
        p = orc.transform_point(q, t, c["point"])
        np.testing.assert_allclose(p, np.array(c["p_read"]), rtol=1e-5, atol=1e-4)
        assert abs(float((q.astype(np.float64) ** 2).sum()) - 1.0) < 1e-6
        assert q[1] == 0 and q[2] == 0     # yaw-only quaternion


# -------------------------------------------------------------- sampling -----
def test_mt19937_known_answer():
    """C++ standard [rand.predef]: 10000th draw of default mt19937 is 4123659995."""
    g = orc.Mt19937()
    v = 0
    for _ in range(10000):
        v = g.next()
    assert v == 4123659995


_CPP = r"""
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>
int main() {
  std::mt19937 gen;  // default seed, as weighted_sampler.h:34
  std::uniform_real_distribution<double> uni{0.0, 1.0};
  std::vector<double> cum;
  double acc = 0;
  for (int i = 0; i < 1000; ++i) { acc += 1.0 + (i % 7) * 0.5; cum.push_back(acc); }
  for (int i = 0; i < 200; ++i) {
    double r = uni(gen);
    auto it = std::upper_bound(cum.begin(), cum.end(), r * cum.back());
    std::printf("%.17g %u\n", r, (unsigned)(it - cum.begin()));
  }
}
"""


def test_weighted_sampler_matches_libstdcxx():
    """weighted_sampler_inl.h:18-28 restated == the same calls on libstdc++."""
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "s.cpp")
        exe = os.path.join(d, "s")
        open(src, "w").write(_CPP)
        subprocess.check_call(["g++", "-O1", "-std=c++14", src, "-o", exe])
        out = subprocess.check_output([exe]).decode().split()
    want_r = np.array(out[0::2], np.float64)
    want_i = np.array(out[1::2], np.int64)
    cum = np.cumsum(1.0 + (np.arange(1000) % 7) * 0.5)
    g1, g2 = orc.Mt19937(), orc.Mt19937()
    got_r = np.array([g1.uniform01() for _ in range(200)])
    got_i = np.array([g2.weighted_draw(cum) for _ in range(200)])
    assert np.array_equal(got_r, want_r)
    assert np.array_equal(got_i, want_i)


# --------------------------------------------------- interpolation basics ----
def _plane_layer(normal, offset, vs=0.1, vps=16, bmin=(-2, -2, -2), bdim=(4, 4, 4)):
    sm = synth.make_submap(synth.plane_sdf(normal, offset), vs, vps, bmin, bdim, trunc=100.0,
                           esdf_max=100.0)
    return sm, orc.Layer(sm.voxel_size, sm.vps, sm.block_index, sm.esdf_distance,
                         sm.esdf_observed)


def test_q_vector_and_neighbour_order_reproduce_trilinear():
    """Neighbour k at base+(k>>2&1,k>>1&1,k&1) and the B_1 table (h:73-81) give the
    ordinary trilinear interpolant; for a linear field it is exact."""
    n = np.array([0.3, -0.5, 0.81], F)
    n /= np.linalg.norm(n)
    sm, layer = _plane_layer(n, 0.123)
    rng = np.random.default_rng(0)
    for p in rng.uniform(-2.9, 2.9, (200, 3)).astype(F):
        ok, d, q = layer.voxels_and_q(p)
        assert ok
        assert (q[1:4] >= 0).all() and (q[1:4] < 1 + 1e-5).all()
        # trilinear weights from (dx,dy,dz)
        dx, dy, dz = q[1], q[2], q[3]
        wts = np.array([(dx if k >> 2 & 1 else 1 - dx) * (dy if k >> 1 & 1 else 1 - dy) *
                        (dz if k & 1 else 1 - dz) for k in range(8)])
        assert abs(float(wts @ d) - float(p @ n - 0.123)) < 2e-5


def test_points_on_voxel_centres_and_negative_blocks():
    n = np.array([0, 0, 1], F)
    sm, layer = _plane_layer(n, 0.0)
    # exactly on a voxel centre (as voxblox computes it) in a negative-index block:
    # centre_offset == 0 is not < 0, so the base voxel is that voxel and Delta == 0
    centres = synth.voxel_centres(sm.voxel_size, sm.vps, sm.block_index)
    b = int(np.where((sm.block_index == (-1, -2, -1)).all(1))[0][0])
    lin = 14 + 16 * (3 + 16 * 15)
    centre = centres[b, lin]
    ok, d, q = layer.voxels_and_q(centre)
    assert ok
    assert d[0] == sm.esdf_distance[b, lin]
    assert np.all(q[1:4] == 0)
    # floor semantics: just below the centre shifts the base one voxel down
    ok2, d2, q2 = layer.voxels_and_q(centre - F(1e-3))
    assert ok2 and d2[0] == sm.esdf_distance[b, 13 + 16 * (2 + 16 * 14)] and q2[3] > 0.98


def test_missing_block_and_unobserved_neighbour():
    sm = synth.make_submap(synth.plane_sdf((0, 0, 1), 0.0), 0.1, 16, (0, 0, 0), (2, 1, 1),
                           trunc=100.0, esdf_max=100.0)
    layer = orc.Layer(sm.voxel_size, sm.vps, sm.block_index, sm.esdf_distance,
                      sm.esdf_observed)
    assert layer.voxels_and_q(np.array([1.0, 0.8, 0.8], F))[0]
    # block containing pos is missing
    assert not layer.voxels_and_q(np.array([-0.3, 0.8, 0.8], F))[0]
    # pos in an allocated block, but a +1 neighbour rolls into a missing block
    assert not layer.voxels_and_q(np.array([3.18, 0.8, 0.8], F))[0]
    # pos in an allocated block whose base voxel shifts into a missing block
    assert not layer.voxels_and_q(np.array([0.02, 0.8, 0.8], F))[0]
    # crossing between the two allocated blocks works
    assert layer.voxels_and_q(np.array([1.6, 0.8, 0.8], F))[0]
    # one unobserved voxel among the 8
    obs = sm.esdf_observed.copy()
    lin = 5 + 16 * (5 + 16 * 5)
    obs[0, lin] = 0
    layer2 = orc.Layer(sm.voxel_size, sm.vps, sm.block_index, sm.esdf_distance, obs)
    assert not layer2.voxels_and_q(np.array([0.5, 0.5, 0.5], F))[0]      # base 4,4,4 -> uses 5,5,5
    assert layer2.voxels_and_q(np.array([0.7, 0.7, 0.7], F))[0]


# ------------------------------------------------------ Evaluate closed form --
def _random_points(rng, n, lo, hi):
    xyz = rng.uniform(lo, hi, (n, 3)).astype(F)
    dist = rng.uniform(-0.3, 0.3, n).astype(F)
    w = rng.uniform(1.5, 10.0, n).astype(F)
    return xyz, dist, w


def test_evaluate_plane_closed_form():
    """Planar SDF => interpolation exact: r_i = (d_i - (n.p' - c)) w_i F and
    J = -w F n^T M with M from the sympy-pinned matrices."""
    n = np.array([0.36, 0.48, 0.8], F)
    sm, layer = _plane_layer(n, 0.2)
    rng = np.random.default_rng(1)
    xyz, dist, w = _random_points(rng, 500, -1.2, 1.2)
    ref_pose = np.array([0.3, -0.2, 0.1, 0.4])
    read_pose = np.array([0.1, 0.25, -0.15, -0.3])
    ok, r, jo, je = orc.reg_evaluate(layer, xyz, dist, w, ref_pose, read_pose)
    assert ok
    q, t = orc.relative_transform(ref_pose, read_pose)
    fac = len(w) / w.astype(np.float64).sum()
    for i in range(0, 500, 7):
        p = orc.transform_point(q, t, xyz[i]).astype(np.float64)
        want = (dist[i] - (p @ n.astype(np.float64) - 0.2)) * w[i] * fac
        assert abs(r[i] - want) < 3e-5 * w[i] * fac
        mo, me = orc.pose_jacobian_matrices(xyz[i, 0], xyz[i, 1], ref_pose, read_pose)
        np.testing.assert_allclose(jo[i], -w[i] * fac * (n @ mo), rtol=2e-3, atol=2e-3)
        np.testing.assert_allclose(je[i], -w[i] * fac * (n @ me), rtol=2e-3, atol=2e-3)


def test_evaluate_jacobian_central_differences():
    """Mirrors the reference's NumericDiff cross-check recipe
    (submap_registration_helper.cpp:50-57) on the config-1 sphere scene."""
    ref, _ = synth.config1_pair()
    layer = orc.Layer(ref.voxel_size, ref.vps, ref.block_index, ref.esdf_distance,
                      ref.esdf_observed)
    xyz, dist, w = orc.find_relevant_voxels(ref.voxel_size, ref.vps, ref.block_index,
                                            ref.tsdf_distance, ref.tsdf_weight,
                                            ref.esdf_distance)
    assert 5000 < len(w) < 100000
    xyz, dist, w = xyz[::17], dist[::17], w[::17]
    ref_pose = np.array([0.05, -0.03, 0.02, 0.02])
    read_pose = np.array([0.0, 0.0, 0.0, 0.0])
    ok, r, jo, je = orc.reg_evaluate(layer, xyz, dist, w, ref_pose, read_pose)
    assert ok
    h = 4e-3
    for blk, jac in ((0, jo), (1, je)):
        for k in range(4):
            poses = [ref_pose.copy(), read_pose.copy()]
            poses[blk][k] += h
            _, rp, _, _ = orc.reg_evaluate(layer, xyz, dist, w, poses[0], poses[1], False)
            poses[blk][k] -= 2 * h
            _, rm, _, _ = orc.reg_evaluate(layer, xyz, dist, w, poses[0], poses[1], False)
            num = (rp - rm) / (2 * h)
            err = np.abs(num - jac[:, k])
            scale = np.abs(jac[:, k]).max() + 1e-9
            # trilinear interpolant is piecewise: allow kinks at a minority of points
            assert np.percentile(err, 80) < 0.03 * scale, (blk, k, np.percentile(err, 80), scale)


def test_evaluate_edge_cases():
    sm, layer = _plane_layer((0, 0, 1), 0.0, bmin=(0, 0, 0), bdim=(1, 1, 1))
    xyz = np.array([[0.8, 0.8, 0.8], [5.0, 5.0, 5.0]], F)     # second: no correspondence
    dist = np.array([0.1, 0.2], F)
    w = np.array([2.0, 3.0], F)
    z = np.zeros(4)
    ok, r, jo, je = orc.reg_evaluate(layer, xyz, dist, w, z, z, no_correspondence_cost=0.25)
    assert ok
    fac = 2 / 5.0
    assert abs(r[1] - 3.0 * 0.25 * fac) < 1e-12                # .cpp:165-166, scaled :275
    assert np.all(jo[1] == 0) and np.all(je[1] == 0)           # .cpp:240-243
    assert abs(r[0] - (0.1 - 0.8) * 2.0 * fac) < 1e-5
    # jacobians == nullptr (.cpp:179) and single null block (.cpp:254,261)
    ok, r2, a, b = orc.reg_evaluate(layer, xyz, dist, w, z, z, want_jac=False)
    assert ok and a is None and b is None and np.array_equal(r, orc.reg_evaluate(
        layer, xyz, dist, w, z, z, no_correspondence_cost=0.25, want_jac=False)[1])
    ok, _, a, b = orc.reg_evaluate(layer, xyz, dist, w, z, z, want_ref=False)
    assert a is None and b is not None
    # sum of weights == 0 -> false (.cpp:273)
    ok, *_ = orc.reg_evaluate(layer, xyz, dist, np.zeros(2, F), z, z)
    assert not ok
    # M_read[:, :3] == -M_ref[:, :3] => J_read[:3] == -J_ref[:3] bit-exactly
    ok, r, jo, je = orc.reg_evaluate(layer, xyz, dist, w, np.array([0.1, 0, 0, 0.3]),
                                     np.array([0, 0.1, 0, -0.2]))
    assert np.array_equal(jo[:, :3], -je[:, :3])


def test_sampling_mode_forces_unit_weight():
    """.cpp:118-122: sampled points get weight 1 => factor 1."""
    sm, layer = _plane_layer((0, 0, 1), 0.0)
    rng = np.random.default_rng(3)
    xyz, dist, w = _random_points(rng, 100, -1, 1)
    cum = np.cumsum(w.astype(np.float64))
    g = orc.Mt19937()
    idx = np.array([g.weighted_draw(cum) for _ in range(int(0.5 * 100))])
    z = np.zeros(4)
    ok, r, jo, je = orc.reg_evaluate(layer, xyz, dist, w, z, z, sample_idx=idx)
    assert ok and len(r) == 50
    ok1, r1, _, _ = orc.reg_evaluate(layer, xyz[idx], dist[idx], np.ones(50, F), z, z)
    np.testing.assert_array_equal(r, r1)


def test_normal_equations_match_materialised():
    ref, _ = synth.config1_pair()
    layer = orc.Layer(ref.voxel_size, ref.vps, ref.block_index, ref.esdf_distance,
                      ref.esdf_observed)
    xyz, dist, w = orc.find_relevant_voxels(ref.voxel_size, ref.vps, ref.block_index,
                                            ref.tsdf_distance, ref.tsdf_weight,
                                            ref.esdf_distance)
    a = np.array([0.05, -0.03, 0.02, 0.02])
    b = np.array([0.0, 0.01, 0.0, -0.01])
    ok, r, jo, je = orc.reg_evaluate(layer, xyz, dist, w, a, b)
    ok2, cost, jtr, jtj = orc.reg_evaluate_normal(layer, xyz, dist, w, a, b)
    assert ok and ok2
    J = np.hstack([jo, je])
    np.testing.assert_allclose(cost, r @ r, rtol=1e-10)
    np.testing.assert_allclose(jtr, J.T @ r, rtol=1e-9, atol=1e-9)
    H = J.T @ J
    np.testing.assert_allclose(jtj, H[np.triu_indices(8)], rtol=1e-9, atol=1e-9)


# ------------------------------------------------------------- extraction ----
def test_find_relevant_voxels_filter_and_layout():
    """voxgraph_submap.cpp:144-201 with the defaults of voxgraph_submap.h:27-28."""
    ref, _ = synth.config1_pair()
    xyz, dist, w = orc.find_relevant_voxels(ref.voxel_size, ref.vps, ref.block_index,
                                            ref.tsdf_distance, ref.tsdf_weight,
                                            ref.esdf_distance)
    mask = (ref.tsdf_weight > 1.0) & (np.abs(ref.tsdf_distance) < 0.3)
    assert len(w) == int(mask.sum())
    centres = synth.voxel_centres(ref.voxel_size, ref.vps, ref.block_index)
    np.testing.assert_array_equal(xyz, centres[mask])       # block order, linear index order
    np.testing.assert_array_equal(dist, ref.esdf_distance[mask])
    np.testing.assert_array_equal(w, ref.tsdf_weight[mask])
    xyz2, dist2, _ = orc.find_relevant_voxels(ref.voxel_size, ref.vps, ref.block_index,
                                              ref.tsdf_distance, ref.tsdf_weight, None)
    np.testing.assert_array_equal(dist2, ref.tsdf_distance[mask])
