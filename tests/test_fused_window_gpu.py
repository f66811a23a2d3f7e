"""The windowed fused kernel (reg_eval_reduce_window_kernel: the reading grid under each chunk of
reference points staged in LDS) against the gathering fused kernel it replaces: the same neighbours, the
same arithmetic, the same order of summation -- so the same BITS, for every pose, point type, voxels-per-
side and no-correspondence cost; and both within 1e-6 of the oracle's sums
(registration_cost_function.cpp:113-291 accumulated as J^T J, J^T r, cost)."""
import os

import numpy as np
import pytest

from oracle import pyoracle as orc
from oracle import synth
from tests import helpers as H

pytestmark = pytest.mark.gpu
GATHER, WINDOW = "622", ("722", "752", "762")


@pytest.fixture(scope="module")
def capi():
    from voxgraph_amd import capi
    return capi


def _normal(batch, poses, variant):
    old = os.environ.get("VGX_FUSED_KERNEL")
    os.environ["VGX_FUSED_KERNEL"] = variant
    try:
        status, normal = batch.evaluate_normal(poses)
    finally:
        if old is None:
            del os.environ["VGX_FUSED_KERNEL"]
        else:
            os.environ["VGX_FUSED_KERNEL"] = old
    return np.array(status), normal.copy()


POSES = [
    ("identity", [[0, 0, 0, 0], [0, 0, 0, 0]]),                      # points ON the reading grid's voxel centres
    ("whole voxels", [[0, 0, 0, 0], [0.2, -0.4, 0.2, 0]]),
    ("small", [[0.02, -0.01, 0.03, 0.01], [0.31, -0.2, 0.08, 0.12]]),
    ("45 deg", [[0, 0, 0, 0.1], [0.5, 0.3, -0.1, 0.1 + np.pi / 4]]),
    ("90 deg", [[0, 0, 0, 0], [0.1, 0.1, 0.0, np.pi / 2]]),
    ("170 deg", [[1, 2, 0, -1.0], [1.2, 1.7, 0.1, -1.0 + 170 * np.pi / 180]]),
    ("half out", [[0, 0, 0, 0], [2.9, -1.7, 0.4, 0.3]]),
    ("far away", [[0, 0, 0, 0], [500.0, 0, 0, 0.3]]),
]


@pytest.mark.parametrize("no_corr", [0.0, 0.25])
def test_same_bits_as_the_gathering_kernel(capi, no_corr):
    ctx = capi.Context(0)
    ref, read = synth.config1_pair(asymmetric=True)
    subs = [H.gpu_submap(capi, ctx, sm, k) for k, sm in enumerate((ref, read))]
    for g in subs:
        g.extract_voxel_points(1.0, 0.3, True)
    cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS, no_correspondence_cost=no_corr)
    pairs = [(0, 1), (1, 0), (0, 0)]
    cfs = [capi.RegistrationCostFunction(ctx, subs[a], subs[b], cfg) for a, b in pairs]
    batch = capi.RegistrationBatch(ctx, cfs, pairs)
    layer = H.oracle_layer(read)
    xyz, dist, w = H.oracle_points(ref)
    for name, poses in POSES:
        poses = np.array(poses, dtype=np.float64)
        st0, n0 = _normal(batch, poses, GATHER)
        for v in WINDOW:
            st, n = _normal(batch, poses, v)
            assert np.array_equal(st, st0), (name, v)
            assert np.array_equal(n, n0), (name, v, np.abs(n - n0).max())
        # ... and they are the oracle's sums (constraint 0: ref -> read)
        ok, cost, jtr, jtj = orc.reg_evaluate_normal(layer, xyz, dist, w, poses[0], poses[1],
                                                     no_correspondence_cost=no_corr)
        want = np.r_[cost, jtr, np.asarray(jtj).reshape(-1)]
        for lo, hi in ((0, 1), (1, 9), (9, 45)):
            scale = np.abs(want[lo:hi]).max()
            if scale > 0:
                assert np.abs(n0[0][lo:hi] - want[lo:hi]).max() <= 1e-6 * scale, (name, lo)
    assert np.abs(n0).sum() >= 0
    for o in [batch] + cfs + subs:
        o.destroy()
    ctx.close()


def test_vps8_isosurface_and_sampled(capi):
    """8 voxels per side; isosurface vertices (not voxel centres; chunks that straddle many blocks); a
    sampling batch (nothing is staged: the kernel must fall back to gathering for every lane)"""
    ctx = capi.Context(0)
    sdf = synth.sphere_ground_sdf((1.6, 1.6, 1.2), 1.0, 0.35)
    sm = synth.make_submap(sdf, 0.1, 8, (0, 0, 0), (4, 4, 3), trunc=0.3, esdf_max=1.0, drop_empty_blocks=True)
    g = H.gpu_submap(capi, ctx, sm, 0)
    g.extract_voxel_points(1.0, 0.3, True)
    g.extract_isosurface_points(1.0)
    poses = np.array([[0.0, 0.0, 0.0, 0.0], [0.13, -0.07, 0.04, 0.6]])
    for kind, ratio in ((capi.POINTS_VOXELS, -1.0), (capi.POINTS_ISOSURFACE, -1.0), (capi.POINTS_VOXELS, 0.3)):
        cfg = capi.default_config(registration_point_type=kind, sampling_ratio=ratio, sampler_seed=7)
        got = []
        for v in (GATHER,) + WINDOW:
            cf = capi.RegistrationCostFunction(ctx, g, g, cfg)     # a fresh engine per variant: the same draws
            batch = capi.RegistrationBatch(ctx, [cf], [(0, 1)])
            got.append(_normal(batch, poses, v)[1])
            batch.destroy()
            cf.destroy()
        assert np.abs(got[0]).sum() > 0
        for n in got[1:]:
            assert np.array_equal(n, got[0]), (kind, ratio)
    g.destroy()
    ctx.close()


def test_full_size_pair(capi):
    """256^3-voxel submaps of the bench's city scene (tiles of 20 chunks, chunks that straddle blocks)"""
    ctx = capi.Context(0)
    bmin, bdim = (-8, -8, -4), (16, 16, 16)
    true = np.array([[0.0, 0.0, 0.0, 0.05], [25.6, 17.0, 0.0, -0.08]])
    subs = [capi.Submap.synth_city(ctx, k, 0.2, 16, bmin, bdim, 0.6, 2.0, 10.0, true[k], 2) for k in range(2)]
    for s in subs:
        assert s.extract_voxel_points(1.0, 0.3, True) > 200_000
        s.release_raw_layers()
    cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS)
    pairs = [(0, 1), (1, 0), (0, 0), (1, 1)]
    cfs = [capi.RegistrationCostFunction(ctx, subs[a], subs[b], cfg) for a, b in pairs]
    batch = capi.RegistrationBatch(ctx, cfs, pairs)
    rng = np.random.default_rng(5)
    for trial in range(4):
        poses = true + rng.normal(0, [0.3, 0.3, 0.1, 0.3], (2, 4))
        if trial == 0:
            poses = np.zeros((2, 4))                                   # self pairs: exact alignment
        st0, n0 = _normal(batch, poses, GATHER)
        assert np.abs(n0).sum() > 0
        for v in WINDOW:
            st, n = _normal(batch, poses, v)
            assert np.array_equal(st, st0) and np.array_equal(n, n0), (trial, v, np.abs(n - n0).max())
    for o in [batch] + cfs + subs:
        o.destroy()
    ctx.close()
