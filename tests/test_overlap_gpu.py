"""Overlap detection on the device vs the numpy restatement of
voxgraph_submap.cpp:245-321 / bounding_box.cpp:28-42 / pose_graph_interface.cpp:109-147."""
import numpy as np
import pytest

from oracle import overlap_oracle as ovl
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


def test_overlapping_pairs_match_reference_logic(capi, ctx):
    sdf = synth.union_sdf(synth.sphere_ground_sdf((1.6, 1.6, 1.2), 1.0, 0.35),
                          synth.sphere_sdf((6.0, 1.0, 1.0), 0.8))
    rng = np.random.default_rng(0)
    poses, subs, gs = [], [], []
    # a chain of partly overlapping submaps plus two far away, some rotated
    layout = [(0, 0, 0, 0.0), (1.6, 0.2, 0.0, 0.3), (3.4, -0.1, 0.05, -0.4), (5.0, 0.3, 0.0, 1.2),
              (20.0, 0, 0, 0.0), (1.0, 3.5, 0.0, 0.7), (-30.0, 4.0, 0.0, -2.0)]
    for i, p in enumerate(layout):
        sm = synth.make_submap(sdf, 0.1, 16, (0, 0, 0), (2, 2, 2), trunc=0.3, pose=p, esdf_max=1.0,
                               drop_empty_blocks=True)
        if sm.n_blocks == 0:
            continue
        g = H.gpu_submap(capi, ctx, sm, i)
        nv, ni = g.extract_voxel_points(), g.extract_isosurface_points()
        vx, _, _ = H.oracle_points(sm)
        ix, _, _ = orc.isosurface_points(sm.voxel_size, 16, sm.block_index, sm.tsdf_distance, sm.tsdf_weight)
        if nv == 0 or ni == 0:
            g.destroy()
            continue
        subs.append(dict(voxel_size=sm.voxel_size, vps=16, block_index=sm.block_index, voxel_xyz=vx, iso_xyz=ix))
        gs.append(g)
        poses.append(np.array(p, np.float64) + np.r_[rng.normal(0, 0.05, 3), rng.normal(0, 0.02)])
    poses = np.array(poses)
    assert len(gs) >= 5
    # surface OBB and mission AABB
    for g, s, p in zip(gs, subs, poses):
        mn, mx = g.surface_obb()
        omn, omx = ovl.surface_obb(s["voxel_xyz"], s["voxel_size"])
        assert np.array_equal(mn, omn) and np.array_equal(mx, omx)
        amn, amx = g.mission_surface_aabb(p)
        bmn, bmx = ovl.mission_aabb(omn, omx, p)
        np.testing.assert_allclose(amn, bmn, rtol=0, atol=1e-5)
        np.testing.assert_allclose(amx, bmx, rtol=0, atol=1e-5)
    got = capi.find_overlapping_pairs(ctx, gs, poses)
    want = ovl.overlapping_pairs(subs, poses)
    print("overlapping pairs:", got)
    assert got == want and 2 <= len(got) < len(gs) * (len(gs) - 1) // 2
    # the pair list feeds the constraint set (pose_graph_interface.cpp:157-174)
    cfg = capi.default_config()
    cfs = [capi.RegistrationCostFunction(ctx, gs[a], gs[b], cfg) for a, b in got]
    batch = capi.RegistrationBatch(ctx, cfs, got)
    status, normal = batch.evaluate_normal(poses)
    assert np.all(status == 0) and np.isfinite(normal).all()
    with pytest.raises(capi.VgxError):
        capi.find_overlapping_pairs(ctx, gs, poses, max_pairs=1)
    for o in [batch] + cfs + gs:
        o.destroy()
