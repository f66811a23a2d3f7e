"""Error conventions of the C ABI on a GPU box: status codes and messages instead of
aborts (the reference CHECK-fails), bounded resources degrade by counting, not crashing."""
import ctypes as C

import numpy as np
import pytest

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


def test_invalid_arguments_are_status_codes(capi, ctx):
    sm, _ = synth.config1_pair()
    with pytest.raises(capi.VgxError) as e:
        capi.Submap(ctx, 0, sm.voxel_size, 4, sm.block_index[:1], None, None,
                    np.zeros((1, 64), F), np.ones((1, 64), np.uint8))
    assert e.value.code == capi.ERR_UNSUPPORTED and "voxels_per_side" in str(e.value)
    with pytest.raises(capi.VgxError) as e:
        capi.Submap(ctx, 0, -1.0, 16, sm.block_index, None, None, sm.esdf_distance, sm.esdf_observed)
    assert e.value.code == capi.ERR_INVALID
    with pytest.raises(capi.VgxError):                  # tsdf distance without weight
        capi.Submap(ctx, 0, sm.voxel_size, 16, sm.block_index, sm.tsdf_distance, None, None, None)
    g = H.gpu_submap(capi, ctx, sm)
    with pytest.raises(capi.VgxError):                  # bad point type
        g.set_points(7, np.zeros((1, 3), F), np.zeros(1, F), np.ones(1, F))
    g.extract_voxel_points()
    # a second context's submap cannot be mixed in
    ctx2 = capi.Context(0)
    g2 = H.gpu_submap(capi, ctx2, sm)
    g2.extract_voxel_points()
    cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS)
    with pytest.raises(capi.VgxError) as e:
        capi.RegistrationCostFunction(ctx, g, g2, cfg)
    assert e.value.code == capi.ERR_INVALID and "another context" in str(e.value)
    # TSDF-distance mode needs a TSDF layer on the reading submap
    only_esdf = capi.Submap(ctx, 5, sm.voxel_size, 16, sm.block_index, None, None,
                            sm.esdf_distance, sm.esdf_observed)
    with pytest.raises(capi.VgxError):
        capi.RegistrationCostFunction(ctx, g, only_esdf, capi.default_config(
            registration_point_type=capi.POINTS_VOXELS, use_esdf_distance=0))
    # sampling constraints batch too (tests/test_batch_sampling_gpu.py); node indices are checked
    cf_s = capi.RegistrationCostFunction(ctx, g, g, capi.default_config(
        registration_point_type=capi.POINTS_VOXELS, sampling_ratio=0.1))
    bs = capi.RegistrationBatch(ctx, [cf_s], [(0, 1)])
    assert bs.num_residuals() == cf_s.num_residuals() > 0
    bs.destroy()
    cf = capi.RegistrationCostFunction(ctx, g, g, cfg)
    batch = capi.RegistrationBatch(ctx, [cf], [(0, 3)])
    with pytest.raises(capi.VgxError):
        batch.evaluate_normal(np.zeros((2, 4)))             # node 3 >= n_nodes
    # NULL residuals
    rc = ctx.lib.vgx_reg_evaluate(cf.h, np.zeros(4).ctypes.data_as(capi.f64p),
                                  np.zeros(4).ctypes.data_as(capi.f64p), None, None, None)
    assert rc == capi.ERR_INVALID
    assert ctx.lib.vgx_reg_num_residuals(None) == -1
    for o in (batch, cf, cf_s, only_esdf, g, g2):
        o.destroy()
    ctx2.close()


def test_empty_submap_and_empty_batch(capi, ctx):
    empty = capi.Submap(ctx, 1, 0.1, 16, np.zeros((0, 3), np.int32), None, None, None, None)
    assert empty.extract_voxel_points() == 0
    sm, _ = synth.config1_pair()
    g = H.gpu_submap(capi, ctx, sm)
    n = g.extract_voxel_points()
    cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS)
    # every point falls outside an empty reading submap: all residuals w * ncc, J == 0
    cf = capi.RegistrationCostFunction(ctx, g, empty, cfg)
    r, jo, je = np.ones(n), np.ones((n, 4)), np.ones((n, 4))
    assert cf.Evaluate([np.zeros(4), np.zeros(4)], r, [jo, je])
    assert np.all(r == 0) and np.all(jo == 0) and np.all(je == 0)
    b = capi.RegistrationBatch(ctx, [], np.zeros((0, 2), np.int32))
    assert b.num_residuals() == 0
    status, normal = b.evaluate_normal(np.zeros((1, 4)))
    assert normal.shape == (0, 45)
    for o in (b, cf, g, empty):
        o.destroy()


def test_tsdf_initial_box_and_pool_are_only_a_reservation(capi, ctx):
    """voxblox::Layer is unbounded.  A box too small for the ray and a pool of one block used to
    COUNT the lost updates; now the layer re-boxes and enlarges its pool before the scan: every
    update lands (64 along the 6 m ray, 34 along the 3 m one) and nothing is dropped."""
    cfg = capi.tsdf_config(default_truncation_distance=0.3, use_const_weight=1, max_ray_length_m=20)
    T = np.array([1, 0, 0, 0, 0.05, 0.05, 0.05], F)
    # box of 2 blocks along x only: the ray leaves it
    layer = capi.TsdfLayer(ctx, 0.1, 16, (0, 0, 0), (2, 1, 1), 8)
    integ = capi.FastTsdfIntegrator(ctx, cfg, layer)
    n = integ.integratePointCloud(T, np.array([[6.0, 0, 0]], F))
    blocks, dropped = layer.stats()
    assert blocks == 4 and n == 64 and dropped == 0 and layer.growths() >= 1
    # big box, pool of 1 block
    layer2 = capi.TsdfLayer(ctx, 0.1, 16, (-2, -2, -2), (8, 4, 4), 1)
    # (a fresh integrator: the first scan's approximate set would make this ray stop early)
    integ2 = capi.FastTsdfIntegrator(ctx, cfg, layer2)
    n2 = integ2.integratePointCloud(T, np.array([[3.0, 0, 0]], F))
    blocks2, dropped2 = layer2.stats()
    assert blocks2 == 3 and n2 == 34 and dropped2 == 0 and layer2.growths() >= 1
    # acknowledging dropped updates (none here) is harmless and leaves the layer usable
    layer2.clear_dropped()
    assert layer2.stats() == (3, 0)
    assert integ2.integratePointCloud(T, np.array([[0.0, 2.5, 0]], F)) > 0 and layer2.stats()[1] == 0
    with pytest.raises(capi.VgxError):
        capi.TsdfLayer(ctx, 0.1, 16, (0, 0, 0), (0, 1, 1), 8)
    with pytest.raises(capi.VgxError):
        capi.TsdfLayer(ctx, 0.1, 16, (0, 0, 0), None, 8)             # a box needs both corners
    for o in (integ, integ2, layer, layer2):
        o.destroy()


def test_create_destroy_cycles_do_not_leak_device_memory(capi, ctx):
    import torch
    sm, _ = synth.config1_pair(asymmetric=True)

    def cycle():
        g = H.gpu_submap(capi, ctx, sm)
        g.generate_esdf()
        g.extract_voxel_points()
        g.extract_isosurface_points()
        cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS)
        cf = capi.RegistrationCostFunction(ctx, g, g, cfg)
        cs = capi.RegistrationCostFunction(ctx, g, g, capi.default_config(sampling_ratio=0.1))
        n = cf.num_residuals()
        r = np.zeros(n)
        cf.Evaluate([np.zeros(4), np.array([0.01, 0, 0, 0.0])], r, None)
        r2 = np.zeros(cs.num_residuals())
        cs.Evaluate([np.zeros(4), np.zeros(4)], r2, None)
        b = capi.RegistrationBatch(ctx, [cf, cf], [(0, 1), (1, 0)])
        b.evaluate_normal(np.zeros((2, 4)))
        pairs = capi.find_overlapping_pairs(ctx, [g, g], np.zeros((2, 4)))
        layer = capi.TsdfLayer(ctx, 0.1, 16, (-2, -2, -2), (4, 4, 4), 64)
        integ = capi.FastTsdfIntegrator(ctx, capi.tsdf_config(), layer)
        integ.integratePointCloud(np.array([1, 0, 0, 0, 0, 0, 0], F), np.array([[1.0, 0.2, 0.1]], F))
        s2 = capi.Submap.from_tsdf_layer(ctx, layer, 3)
        for o in (s2, integ, layer, b, cs, cf, g):
            o.destroy()
        return pairs

    cycle()
    ctx.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(10):
        cycle()
    ctx.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 8 << 20, (free0, free1)


def test_large_sparse_block_extent_and_table_limit(capi, ctx):
    """A submap whose blocks span a huge box still works through the dense block table up
    to 2^28 cells; beyond that the upload is refused with UNSUPPORTED (no silent hashing)."""
    plane = synth.plane_sdf((0, 0, 1), 0.05)
    sm = synth.make_submap(plane, 0.1, 16, (0, 0, -1), (2, 1, 2), trunc=100.0, esdf_max=100.0)
    # move the second x-block 500 blocks away (800 m): 501 x 1 x 2 table cells, both blocks usable
    bi = sm.block_index.copy()
    far = bi[:, 0] == 1
    bi[far, 0] = 500
    # the ESDF values of the moved blocks no longer match their position; re-evaluate the plane
    centres = synth.voxel_centres(sm.voxel_size, 16, bi)
    ed = (centres[..., 2] - np.float32(0.05)).astype(F)
    g = capi.Submap(ctx, 0, sm.voxel_size, 16, bi, None, None, ed, np.ones_like(sm.esdf_observed))
    xyz = np.array([[0.8, 0.8, 0.2], [800.8, 0.8, -0.3], [400.0, 0.8, 0.0]], F)   # near, far, gap
    g.set_points(capi.POINTS_ISOSURFACE, xyz, np.zeros(3, F), np.ones(3, F))
    cf = capi.RegistrationCostFunction(ctx, g, g, capi.default_config())
    r, jo, je = np.zeros(3), np.zeros((3, 4)), np.zeros((3, 4))
    assert cf.Evaluate([np.zeros(4), np.zeros(4)], r, [jo, je])
    np.testing.assert_allclose(r[:2], [-(0.2 - 0.05), -(-0.3 - 0.05)], atol=1e-5)
    assert r[2] == 0 and np.all(jo[2] == 0)                     # in the gap: no block, no correspondence
    np.testing.assert_allclose(jo[:2, 2], [-1.0, -1.0], atol=1e-4)
    cf.destroy()
    g.destroy()
    bi2 = np.array([[0, 0, 0], [1 << 12, 1 << 12, 1 << 6]], np.int32)    # 2^30 cells
    with pytest.raises(capi.VgxError) as e:
        capi.Submap(ctx, 1, 0.1, 16, bi2, None, None, np.zeros((2, 4096), F), np.ones((2, 4096), np.uint8))
    assert e.value.code == capi.ERR_UNSUPPORTED


def test_round2_entry_points_reject_bad_arguments(capi, ctx):
    """status codes, never a crash: layer upload / reserve, merged integrator, multi-context batch"""
    import ctypes as C
    lib = ctx.lib
    layer = capi.TsdfLayer(ctx, 0.1, 16)
    # upload: NULL arrays with n > 0, negative n
    assert lib.vgx_tsdf_layer_upload(layer.h, 3, None, None, None, None) == capi.ERR_INVALID
    assert lib.vgx_tsdf_layer_upload(layer.h, -1, None, None, None, None) == capi.ERR_INVALID
    assert lib.vgx_tsdf_layer_upload(None, 0, None, None, None, None) == capi.ERR_INVALID
    # an empty upload is legal and leaves an empty layer
    layer.upload(np.zeros((0, 3), np.int32), np.zeros((0, 4096), F), np.zeros((0, 4096), F))
    assert layer.stats() == (0, 0)
    # reserve: NULL origin, negative / NaN reach
    assert lib.vgx_tsdf_layer_reserve(layer.h, None, C.c_float(1.0)) == capi.ERR_INVALID
    o = np.zeros(3, F)
    assert lib.vgx_tsdf_layer_reserve(layer.h, o.ctypes.data_as(capi.f32p), C.c_float(-1.0)) == capi.ERR_INVALID
    assert lib.vgx_tsdf_layer_reserve(layer.h, o.ctypes.data_as(capi.f32p), C.c_float(float("nan"))) == capi.ERR_INVALID
    layer.reserve(o, 3.0)
    assert layer.growths() >= 1 and layer.stats() == (0, 0)
    # merged integrator: NULL points with n > 0, negative n, empty scan
    integ = capi.FastTsdfIntegrator(ctx, capi.tsdf_config(), layer)
    T = np.array([1, 0, 0, 0, 0, 0, 0], F)
    assert lib.vgx_tsdf_integrate_merged(integ.h, T.ctypes.data_as(capi.f32p), None, None, 5, 0, None) == capi.ERR_INVALID
    assert lib.vgx_tsdf_integrate_merged(integ.h, T.ctypes.data_as(capi.f32p), None, None, -1, 0, None) == capi.ERR_INVALID
    assert integ.integratePointCloudMerged(T, np.zeros((0, 3), F)) == 0
    # every point invalid (below the minimum ray length): nothing is integrated, nothing breaks
    assert integ.integratePointCloudMerged(T, np.full((100, 3), 0.001, F)) == 0 and layer.stats() == (0, 0)
    # multi-context batch: no contexts, too many, NULL context
    h = capi.vp()
    assert lib.vgx_reg_multi_create(0, None, 0, None, None, C.byref(h)) == capi.ERR_INVALID
    arr = (capi.vp * 17)(*[ctx.h] * 17)
    assert lib.vgx_reg_multi_create(17, arr, 0, None, None, C.byref(h)) == capi.ERR_INVALID
    arr2 = (capi.vp * 2)(ctx.h, None)
    assert lib.vgx_reg_multi_create(2, arr2, 0, None, None, C.byref(h)) == capi.ERR_INVALID
    # an empty multi batch evaluates to an all-zero buffer
    empty = capi.RegistrationMulti([ctx], [], np.zeros((0, 2), np.int32))
    fused, _ = empty.evaluate_fused(np.zeros((2, 4)))
    assert fused.shape == (capi.fused_size(2, 0),) and not fused.any()
    empty.destroy()
    assert capi.lpt_shards([], 3).shape == (0,)
    for obj in (integ, layer):
        obj.destroy()


def test_round6_entry_points_reject_bad_arguments(capi, ctx):
    """status codes, never a crash: f64 rows, rows kept by the batch + fetch, library-allocated outputs, the void TSDF call;
    an EMPTY batch goes through every one of them"""
    import ctypes as C
    lib = ctx.lib
    poses = np.zeros((2, 4))
    pp = poses.ctypes.data_as(capi.f64p)
    empty = capi.RegistrationBatch(ctx, [], np.zeros((0, 2), np.int32))
    r, jo, je, ms = empty.alloc_outputs(poses, n_candidates=2)          # nothing to time: the first set, untimed
    assert r and jo and je and ms == 0.0
    assert empty.evaluate_points_f64(poses, r, jo, je).shape == (0,)
    empty.free_outputs(r, jo, je)
    assert empty.evaluate_rows_f64(poses).shape == (0,)
    assert lib.vgx_reg_batch_fetch_rows_f64(empty.h, 0, None, None, None) == capi.ERR_INVALID      # no such constraint
    out = [capi.vp(), capi.vp(), capi.vp()]
    for bad_n in (0, 17, -1):
        assert lib.vgx_reg_batch_alloc_outputs(empty.h, pp, 2, bad_n, 1, 1, C.byref(out[0]), C.byref(out[1]), C.byref(out[2]), None) == capi.ERR_INVALID
    assert lib.vgx_reg_batch_alloc_outputs(empty.h, pp, 2, 2, 1, 1, C.byref(out[0]), None, C.byref(out[2]), None) == capi.ERR_INVALID   # wanted, nowhere to put it
    assert lib.vgx_reg_batch_alloc_outputs(None, pp, 2, 2, 0, 0, C.byref(out[0]), None, None, None) == capi.ERR_INVALID
    assert lib.vgx_reg_batch_evaluate_points_f64(empty.h, pp, 2, None, None, None, None) == capi.ERR_INVALID                           # residuals == NULL
    assert lib.vgx_reg_batch_evaluate_rows_f64(None, pp, 2, 1, 1, None) == capi.ERR_INVALID
    assert lib.vgx_reg_batch_evaluate_rows_f64(empty.h, None, 2, 1, 1, None) == capi.ERR_INVALID
    assert lib.vgx_reg_batch_free_outputs(None, None, None, None) == capi.ERR_INVALID
    empty.destroy()
    # the void TSDF call (n_updates == NULL): NULL points with n > 0 and a negative n are refused, an empty scan is legal
    layer = capi.TsdfLayer(ctx, 0.1, 16)
    integ = capi.FastTsdfIntegrator(ctx, capi.tsdf_config(), layer)
    T = np.array([1, 0, 0, 0, 0, 0, 0], F)
    assert lib.vgx_tsdf_integrate(integ.h, T.ctypes.data_as(capi.f32p), None, None, 5, 0, None) == capi.ERR_INVALID
    assert lib.vgx_tsdf_integrate(integ.h, T.ctypes.data_as(capi.f32p), None, None, -1, 0, None) == capi.ERR_INVALID
    assert integ.integratePointCloud(T, np.zeros((0, 3), F), count=False) == 0
    assert integ.integratePointCloud(T, np.full((100, 3), 0.001, F), count=False) == 0 and layer.stats() == (0, 0)
    for obj in (integ, layer):
        obj.destroy()
