"""Pins what the ESDF restatement (oracle/esdf_oracle.c, voxblox [recalled]) does."""
import numpy as np

from oracle import pyoracle as orc
from oracle import synth

F = np.float32


def _plane_tsdf(vs=0.1, trunc=0.3, normal=(0, 0, 1), offset=0.0, bdim=(3, 3, 3), observed_within=10.0):
    sm = synth.make_submap(synth.plane_sdf(normal, offset), vs, 16, (-1, -1, -1), bdim, trunc=trunc,
                           esdf_max=2.0)
    sm.tsdf_weight[:] = 10.0
    return sm


def test_axis_aligned_plane_gives_exact_distances_up_to_max():
    """For a wall normal to an axis the 26-neighbour wavefront walks straight: ESDF = true
    distance (quasi-Euclidean = Euclidean along axes), capped at default_distance."""
    sm = _plane_tsdf()
    ed, eo, n = orc.esdf_from_tsdf(sm.voxel_size, 16, sm.block_index, sm.tsdf_distance, sm.tsdf_weight)
    assert eo.all() and n > 0
    z = synth.voxel_centres(sm.voxel_size, 16, sm.block_index)[..., 2]
    want = np.clip(z, -2.0, 2.0)
    # fixed band copies the TSDF (|d| < 0.2); beyond it steps of exactly one voxel accumulate
    fixed = np.abs(sm.tsdf_distance) < 0.2
    assert np.array_equal(ed[fixed], sm.tsdf_distance[fixed])
    inside = np.abs(z) < 1.9
    assert np.abs(ed - want)[inside].max() < 2e-3 + 1e-3 * 20


def test_oblique_plane_is_quasi_euclidean_overestimate():
    n = np.array([0.6, 0.0, 0.8], F)
    sm = _plane_tsdf(normal=n, offset=0.1)
    ed, eo, _ = orc.esdf_from_tsdf(sm.voxel_size, 16, sm.block_index, sm.tsdf_distance, sm.tsdf_weight)
    c = synth.voxel_centres(sm.voxel_size, 16, sm.block_index)
    true = c @ n - 0.1
    sel = (np.abs(true) > 0.3) & (np.abs(true) < 1.2)
    err = np.abs(ed) - np.abs(true)
    assert err[sel].min() > -0.03          # never much shorter than the true distance
    assert err[sel].max() < 0.25 * np.abs(true[sel]).max()   # chamfer error of the 26-neighbourhood
    assert np.all(np.sign(ed[sel]) == np.sign(true[sel]))


def test_unobserved_voxels_block_the_wavefront_and_stay_unobserved():
    sm = _plane_tsdf()
    z = synth.voxel_centres(sm.voxel_size, 16, sm.block_index)[..., 2]
    sm.tsdf_weight[(z > 0.5) & (z < 0.7)] = 0.0          # an unobserved slab
    ed, eo, _ = orc.esdf_from_tsdf(sm.voxel_size, 16, sm.block_index, sm.tsdf_distance, sm.tsdf_weight)
    assert not eo[(z > 0.5) & (z < 0.7)].any()
    above = z > 0.7
    assert eo[above].all() and np.all(ed[above] == F(2.0))   # never reached: default distance
    below = (z > 0.25) & (z < 0.5)
    assert np.abs(ed[below] - z[below]).max() < 0.02
