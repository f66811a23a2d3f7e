"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np

from oracle import pyoracle as orc

# north_star tolerance: 1e-4 relative on residuals/Jacobians.  Residuals
# legitimately cross zero, so "relative" is against max(|oracle|, FLOOR * ||oracle||_inf)
# (SURVEY.md 4.3).
REL_TOL = 1e-4
FLOOR = 1e-3


def assert_parity(got, want, what="", rel=REL_TOL, floor=FLOOR):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    if want.size == 0:
        return 0.0
    scale = np.maximum(np.abs(want), floor * np.abs(want).max())
    scale = np.maximum(scale, 1e-300)
    err = np.abs(got - want) / scale
    worst = float(err.max())
    assert worst <= rel, f"{what}: worst relative error {worst:.3e} at {int(err.argmax())}"
    return worst


def oracle_layer(sm, use_esdf=True):
    if use_esdf:
        return orc.Layer(sm.voxel_size, sm.vps, sm.block_index, sm.esdf_distance, sm.esdf_observed)
    return orc.Layer(sm.voxel_size, sm.vps, sm.block_index, sm.tsdf_distance,
                     (sm.tsdf_weight > 0).astype(np.uint8))


def oracle_points(sm, use_esdf=True, min_w=1.0, max_d=0.3):
    return orc.find_relevant_voxels(sm.voxel_size, sm.vps, sm.block_index, sm.tsdf_distance,
                                    sm.tsdf_weight, sm.esdf_distance if use_esdf else None,
                                    min_w, max_d)


def gpu_submap(capi, ctx, sm, submap_id=0):
    return capi.Submap(ctx, submap_id, sm.voxel_size, sm.vps, sm.block_index, sm.tsdf_distance,
                       sm.tsdf_weight, sm.esdf_distance, sm.esdf_observed)


# the reference's perturbation grid (config/registration_test_bench.yaml:9-13),
# scaled to the voxel size as in SURVEY.md 8d config 1
def test_bench_grid(voxel_size):
    s = voxel_size / 0.2
    out = []
    for x in (-0.6 * s, 0.0, 0.3 * s):
        for y in (-0.6 * s, 0.0, 0.3 * s):
            for z in (-0.6 * s, 0.0, 0.3 * s):
                for yaw in (-0.2, 0.0, 0.1):
                    out.append(np.array([x, y, z, yaw]))
    return out
