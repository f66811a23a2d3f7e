"""Pins the callers' side of REG to the REFERENCE'S OWN SOURCE:
voxgraph_submap.cpp (finishSubmap -> findRelevantVoxelIndices, findIsosurfaceVertices,
getSubmapFrameSurfaceObb, overlapsWith) and bounding_box.cpp, compiled from /root/reference into
oracle/_ref/libref_reg.so against the stand-in headers of oracle/ref_shims.

Pinned here: the voxel filter and point assembly, which distance each point carries, the surface
OBB, the mission-frame AABB, the isosurface block set and the two-stage overlap test.  NOT pinned
(shim = the same recalled restatement): voxblox's marching-cubes vertex positions and
getConnectedMesh (see oracle/ref_shims/voxblox/mesh/mesh_integrator.h), the interpolator, minkindr.

Live tests: skipped where the library was not built.  The golden fixture
tests/golden/ref_submap_scene.npz carries the same outputs to the GPU box."""
import os

import numpy as np
import pytest

from oracle import overlap_oracle as ovl
from oracle import pyoracle as orc
from oracle import ref_reg, synth

F = np.float32
pytestmark = pytest.mark.skipif(not ref_reg.available(),
                                reason="oracle/_ref/libref_reg.so not built (needs /root/reference)")


def reorder_blocks(sm, order):
    """arrays of `sm` permuted into the given block order"""
    lut = {tuple(b): i for i, b in enumerate(np.asarray(sm.block_index).tolist())}
    perm = np.array([lut[tuple(b)] for b in order.tolist()])
    nv = sm.vps ** 3
    def pick(a):
        return np.ascontiguousarray(np.asarray(a).reshape(-1, nv)[perm])
    return order.copy(), pick(sm.tsdf_distance), pick(sm.tsdf_weight), pick(sm.esdf_distance), pick(sm.esdf_observed)


def scene_submaps():
    sdf = synth.union_sdf(synth.sphere_ground_sdf((1.6, 1.6, 1.2), 1.0, 0.35),
                          synth.sphere_sdf((6.0, 1.0, 1.0), 0.8))
    layout = [(0, 0, 0, 0.0), (1.6, 0.2, 0.0, 0.3), (3.4, -0.1, 0.05, -0.4), (5.0, 0.3, 0.0, 1.2),
              (20.0, 0, 0, 0.0), (1.0, 3.5, 0.0, 0.7), (-30.0, 4.0, 0.0, -2.0)]
    rng = np.random.default_rng(0)
    out = []
    for i, p in enumerate(layout):
        sm = synth.make_submap(sdf, 0.1, 16, (0, 0, 0), (2, 2, 2), trunc=0.3, pose=p, esdf_max=1.0,
                               drop_empty_blocks=True)
        pose = np.array(p, np.float64) + np.r_[rng.normal(0, 0.05, 3), rng.normal(0, 0.02)]
        if sm.n_blocks:
            out.append((i, sm, pose))
    return out


@pytest.mark.parametrize("use_esdf,min_w,max_d", [(True, 1.0, 0.3), (False, 1.0, 0.3), (True, 5.0, 0.12)])
def test_find_relevant_voxel_indices(use_esdf, min_w, max_d):
    """voxgraph_submap.cpp:144-201: same voxels, same order, same position/distance/weight bits"""
    ref, _ = synth.config1_pair(seed=0, asymmetric=True)
    R = ref_reg.Submap(0, ref.pose, ref.voxel_size, ref.vps, ref.block_index, ref.tsdf_distance,
                       ref.tsdf_weight, ref.esdf_distance, ref.esdf_observed, min_w, max_d, use_esdf)
    order = R.block_order()
    assert sorted(map(tuple, order.tolist())) == sorted(map(tuple, np.asarray(ref.block_index).tolist()))
    bi, td, tw, ed, eo = reorder_blocks(ref, order)
    xyz, d, w = orc.find_relevant_voxels(ref.voxel_size, ref.vps, bi, td, tw, ed if use_esdf else None, min_w, max_d)
    rx, rd, rw = R.points(ref_reg.POINTS_VOXELS)
    assert len(rw) == len(w) > 1000
    assert np.array_equal(rx, xyz) and np.array_equal(rd, d) and np.array_equal(rw, w)


def test_find_isosurface_vertices_glue():
    """voxgraph_submap.cpp:203-243 around the mesh stand-in: vertices carry the INTERPOLATED TSDF
    distance and weight, the CHECK_LE(distance, 1e-2 voxel) holds, and the isosurface block set is
    the set of blocks containing vertices"""
    ref, _ = synth.config1_pair(seed=0, asymmetric=True)
    R = ref_reg.Submap(0, ref.pose, ref.voxel_size, ref.vps, ref.block_index, ref.tsdf_distance,
                       ref.tsdf_weight, ref.esdf_distance, ref.esdf_observed)
    bi, td, tw, _, _ = reorder_blocks(ref, R.block_order())
    xyz, d, w = orc.isosurface_points(ref.voxel_size, ref.vps, bi, td, tw, 1.0)
    rx, rd, rw = R.points(ref_reg.POINTS_ISOSURFACE)
    assert len(rw) == len(w) > 500
    assert np.array_equal(rx, xyz) and np.array_equal(rd, d) and np.array_equal(rw, w)
    want_blocks = ovl.isosurface_blocks(xyz, ref.voxel_size, ref.vps)
    got_blocks = np.unique(R.isosurface_blocks().astype(np.int64), axis=0)
    assert np.array_equal(got_blocks, want_blocks)


def test_surface_obb_aabb_and_overlap_list():
    """voxgraph_submap.cpp:245-321,379-383 + bounding_box.cpp:12-42 + the i<j pair loop of
    pose_graph_interface.cpp:109-147"""
    subs = scene_submaps()
    refs, dicts, poses = [], [], []
    for i, sm, pose in subs:
        R = ref_reg.Submap(i, pose, sm.voxel_size, sm.vps, sm.block_index, sm.tsdf_distance, sm.tsdf_weight,
                           sm.esdf_distance, sm.esdf_observed)
        bi, td, tw, ed, _ = reorder_blocks(sm, R.block_order())
        vx, _, _ = orc.find_relevant_voxels(sm.voxel_size, sm.vps, bi, td, tw, ed, 1.0, 0.3)
        ix, _, _ = orc.isosurface_points(sm.voxel_size, sm.vps, bi, td, tw, 1.0)
        if len(vx) == 0 or len(ix) == 0:
            continue
        omn, omx = ovl.surface_obb(vx, sm.voxel_size)
        rmn, rmx = R.surface_obb()
        assert np.array_equal(rmn, omn) and np.array_equal(rmx, omx)
        amn, amx = ovl.mission_aabb(omn, omx, pose)
        bmn, bmx = R.mission_surface_aabb()
        assert np.array_equal(amn, bmn) and np.array_equal(amx, bmx)
        refs.append(R)
        poses.append(pose)
        dicts.append(dict(voxel_size=sm.voxel_size, vps=sm.vps, block_index=sm.block_index, voxel_xyz=vx, iso_xyz=ix))
    assert len(refs) >= 5
    want = ovl.overlapping_pairs(dicts, poses)
    got = [(a, b) for a in range(len(refs)) for b in range(a + 1, len(refs)) if refs[a].overlapsWith(refs[b])]
    assert got == want and 2 <= len(got) < len(refs) * (len(refs) - 1) // 2
    # and under many random re-posings (AABB early-outs, rotated boxes, block probes)
    rng = np.random.default_rng(3)
    n_true = 0
    for _ in range(150):
        a, b = rng.choice(len(refs), 2, replace=False)
        pa = np.r_[rng.uniform(-4, 4, 2), rng.uniform(-0.5, 0.5), rng.uniform(-3.1, 3.1)]
        pb = np.r_[rng.uniform(-4, 4, 2), rng.uniform(-0.5, 0.5), rng.uniform(-3.1, 3.1)]
        refs[a].set_pose(pa)
        refs[b].set_pose(pb)
        r = refs[a].overlapsWith(refs[b])
        assert r == ovl.overlaps_with(dicts[a], pa, dicts[b], pb)
        n_true += r
    assert 10 < n_true < 140
