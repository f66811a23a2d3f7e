"""Pins the isosurface-point restatement (oracle/iso_oracle.c) to known geometry."""
import numpy as np

from oracle import pyoracle as orc
from oracle import synth

F = np.float32


def test_plane_isosurface_vertices_lie_on_the_plane_one_per_half_voxel_cell():
    n = np.array([0.0, 0.6, 0.8], F)
    sm = synth.make_submap(synth.plane_sdf(n, 0.13), 0.1, 16, (-1, -1, -1), (2, 2, 2), trunc=0.3)
    xyz, d, w = orc.isosurface_points(sm.voxel_size, 16, sm.block_index, sm.tsdf_distance, sm.tsdf_weight)
    assert len(w) > 500
    # vertices are zero crossings of a linear field along grid edges: exactly on the plane
    assert np.abs(xyz @ n - 0.13).max() < 2e-5
    # interpolated TSDF distance ~ 0 (the reference CHECKs <= 1e-2 voxel, voxgraph_submap.cpp:230)
    assert np.abs(d).max() < 1e-2 * 0.1 and np.all(w == 10.0)
    # connected mesh: no two vertices share a 0.5-voxel cell
    cells = np.round(xyz.astype(np.float64) / np.float64(F(0.05)))
    assert len(np.unique(cells, axis=0)) == len(cells)
    # every vertex sits on a grid edge: two coordinates are voxel centres
    frac = np.abs(((xyz / F(0.1)) - 0.5) - np.round((xyz / F(0.1)) - 0.5))
    assert np.all((frac < 1e-3).sum(1) >= 2)


def test_sphere_surface_and_unobserved_cells():
    sm, _ = synth.config1_pair()
    xyz, d, w = orc.isosurface_points(sm.voxel_size, 16, sm.block_index, sm.tsdf_distance, sm.tsdf_weight)
    r = np.linalg.norm(xyz - np.array([3.2, 3.2, 3.2], F), axis=1)
    on_sphere = np.abs(r - 2.0) < 0.02
    on_ground = np.abs(xyz[:, 2] - 0.45) < 0.02
    assert (on_sphere | on_ground).all() and on_sphere.sum() > 2000 and on_ground.sum() > 1000
    # min_weight above the TSDF weight: nothing is observed enough, no points
    x2, _, _ = orc.isosurface_points(sm.voxel_size, 16, sm.block_index, sm.tsdf_distance,
                                     sm.tsdf_weight, min_weight=10.0)
    assert len(x2) == 0
    # knock out weights in a slab: no vertices there
    tw = sm.tsdf_weight.copy()
    c = synth.voxel_centres(sm.voxel_size, 16, sm.block_index)
    tw[(c[..., 0] > 3.0) & (c[..., 0] < 3.6)] = 0.0
    x3, _, _ = orc.isosurface_points(sm.voxel_size, 16, sm.block_index, sm.tsdf_distance, tw)
    assert not ((x3[:, 0] > 3.0) & (x3[:, 0] < 3.6)).any() and 0 < len(x3) < len(xyz)
