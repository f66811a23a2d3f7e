"""Device extraction of kIsosurfacePoints (voxgraph_submap.cpp:203-243, the registration
points of the shipped "explicit_to_implicit" method) vs the CPU restatement: bit-exact,
same order."""
import numpy as np
import pytest

from oracle import pyoracle as orc
from oracle import synth
from tests import helpers as H

pytestmark = pytest.mark.gpu
F = np.float32


@pytest.fixture(scope="module")
def capi():
    from voxgraph_amd import capi
    capi.load()
    return capi


@pytest.fixture(scope="module")
def ctx(capi):
    c = capi.Context(0)
    yield c
    c.close()


def _compare(capi, ctx, sm, min_w=1.0):
    g = H.gpu_submap(capi, ctx, sm)
    n = g.extract_isosurface_points(min_w)
    xyz, d, w = orc.isosurface_points(sm.voxel_size, sm.vps, sm.block_index, sm.tsdf_distance,
                                      sm.tsdf_weight, min_w)
    assert n == len(w), (n, len(w))
    gx, gd, gw = g.download_points(capi.POINTS_ISOSURFACE)
    assert np.array_equal(gx, xyz) and np.array_equal(gd, d) and np.array_equal(gw, w)
    return g, (xyz, d, w)


def test_isosurface_points_bit_exact_on_dense_and_sparse_scenes(capi, ctx):
    sm, _ = synth.config1_pair(asymmetric=True)
    g, (xyz, d, w) = _compare(capi, ctx, sm)
    assert len(w) > 5000
    g.destroy()
    # sparse blocks, negative block indices, varying weights (interpolated weight differs per vertex)
    sdf = synth.sphere_ground_sdf((0.3, -0.2, 0.1), 1.5, -1.0)
    sp = synth.make_submap(sdf, 0.1, 16, (-2, -2, -2), (4, 4, 4), trunc=0.3, esdf_max=0.8,
                           drop_empty_blocks=True)
    rng = np.random.default_rng(0)
    sp.tsdf_weight[:] = np.where(sp.tsdf_weight > 0, rng.uniform(0.5, 12.0, sp.tsdf_weight.shape), 0).astype(F)
    g, (xyz, d, w) = _compare(capi, ctx, sp)
    assert len(w) > 1000 and len(np.unique(w)) > 100
    g.destroy()
    g, _ = _compare(capi, ctx, sp, min_w=6.0)       # many cells fail the min-weight test
    g.destroy()
    # vps = 8
    s8 = synth.make_submap(synth.sphere_ground_sdf((1.0, 1.0, 1.0), 0.7, 0.3), 0.05, 8, (0, 0, 0),
                           (5, 5, 5), trunc=0.15, esdf_max=0.5)
    g, (xyz, _, _) = _compare(capi, ctx, s8)
    assert len(xyz) > 1000
    g.destroy()


def test_registration_with_device_isosurface_points_default_config(capi, ctx):
    """RegistrationCostFunction::Config defaults (kIsosurfacePoints, ESDF distance) with the
    points produced on the device."""
    sm, _ = synth.config1_pair(asymmetric=True)
    g = H.gpu_submap(capi, ctx, sm)
    n = g.extract_isosurface_points()
    xyz, d, w = orc.isosurface_points(sm.voxel_size, sm.vps, sm.block_index, sm.tsdf_distance, sm.tsdf_weight)
    cf = capi.RegistrationCostFunction(ctx, g, g, capi.default_config())      # reference defaults
    assert cf.num_residuals() == n
    a, b = np.array([0.04, -0.02, 0.03, 0.02]), np.array([0.0, 0.01, 0.0, -0.01])
    r, jo, je = np.zeros(n), np.zeros((n, 4)), np.zeros((n, 4))
    assert cf.Evaluate([a, b], r, [jo, je])
    ok, r0, jo0, je0 = orc.reg_evaluate(H.oracle_layer(sm), xyz, d, w, a, b)
    H.assert_parity(r, r0, "residual")
    H.assert_parity(jo, jo0, "jac_ref")
    H.assert_parity(je, je0, "jac_read")
    # mirrored constraint for isosurface points (pose_graph.cpp:62-71) is just the swapped pair
    cf2 = capi.RegistrationCostFunction(ctx, g, g, capi.default_config(sampling_ratio=0.05))
    assert cf2.num_residuals() == int(np.float32(0.05) * np.float32(n))       # shipped yaml :34
    r2 = np.zeros(cf2.num_residuals())
    assert cf2.Evaluate([a, b], r2, None)
    for o in (cf, cf2, g):
        o.destroy()


def test_isosurface_fullsize_256(capi, ctx):
    g = capi.Submap.synth_city(ctx, 0, 0.2, 16, (-8, -8, -4), (16, 16, 16), 0.6, 2.0, 10.0,
                               np.array([3.0, -2.0, 0.0, 0.1]), 2)
    ctx.timer_start()
    n = g.extract_isosurface_points()
    ms = ctx.timer_stop()
    print(f"isosurface points 256^3: {n} points, {ms:.2f} ms")
    xyz, d, w = g.download_points(capi.POINTS_ISOSURFACE)
    assert n > 100_000 and np.abs(d).max() < 1e-2 * 0.2 * 5       # near the zero level set
    cells = np.round(xyz.astype(np.float64) / np.float64(np.float32(0.1)))
    assert len(np.unique(cells, axis=0)) == n                     # connected-mesh property
    g.destroy()
