"""Miniature of BASELINE config 2 (sensor -> submaps -> pose graph), both hot paths
chained on the GPU the way voxgraph chains them:
  scans --TSDF kernel--> active layers --finishSubmap--> finished submaps
        --device point extraction--> registration constraints --REG kernels--> solve.
The finished submap never leaves the device: vgx_submap_from_tsdf_layer ->
vgx_submap_generate_esdf -> vgx_submap_extract_voxel_points."""
import numpy as np
import pytest

from harness import lm
from harness.backends import GpuBackend

pytestmark = pytest.mark.gpu
F = np.float32
VS, VPS, TRUNC = 0.1, 16, 0.3


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


def _cast(origin, dirs):
    """first hit of rays with a 6 x 5 x 3 m room containing an off-centre box; world frame"""
    lo, hi = np.array([-3.0, -2.5, -1.0]), np.array([3.0, 2.5, 2.0])
    with np.errstate(divide="ignore", invalid="ignore"):
        t_room = np.where(dirs > 0, (hi - origin) / dirs, np.where(dirs < 0, (lo - origin) / dirs, np.inf)).min(1)
        blo, bhi = np.array([1.0, 0.2, -1.0]), np.array([2.0, 1.4, 0.6])
        t1, t2 = (blo - origin) / dirs, (bhi - origin) / dirs
    tn, tf = np.minimum(t1, t2).max(1), np.maximum(t1, t2).min(1)
    hit_box = (tn < tf) & (tn > 0)
    return np.where(hit_box, np.minimum(tn, t_room), t_room)


def _yaw_q(yaw):
    return np.array([np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)])


def _build_submap(capi, ctx, submap_pose, sensor_poses, submap_id):
    """integrate scans taken at world sensor poses into a layer expressed in the submap frame"""
    az = np.linspace(-np.pi, np.pi, 720, endpoint=False)
    el = np.linspace(-0.6, 0.6, 48)
    A, E = np.meshgrid(az, el)
    d_s = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], -1).reshape(-1, 3)
    layer = capi.TsdfLayer(ctx, VS, VPS, (-4, -4, -2), (8, 8, 4), 256)
    integ = capi.FastTsdfIntegrator(ctx, capi.tsdf_config(default_truncation_distance=TRUNC,
                                                          max_ray_length_m=10.0, use_const_weight=1,
                                                          deterministic=1), layer)   # reproducible: no race can fail this
    cs, ss = np.cos(submap_pose[3]), np.sin(submap_pose[3])
    R_ws = np.array([[cs, -ss, 0], [ss, cs, 0], [0, 0, 1.0]])
    for (x, y, z, yaw) in sensor_poses:
        c, s = np.cos(yaw), np.sin(yaw)
        R_wc = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
        origin = np.array([x, y, z])
        t = _cast(origin, d_s @ R_wc.T)
        pts_c = (d_s * t[:, None]).astype(F)
        # T_submap_sensor = T_world_submap^-1 * T_world_sensor
        t_sc = R_ws.T @ (origin - submap_pose[:3])
        T = np.concatenate([_yaw_q(yaw - submap_pose[3]), t_sc]).astype(F)
        integ.integratePointCloud(T, pts_c)
    assert layer.stats()[1] == 0
    # finishSubmap() on the device (voxgraph_submap.cpp:84-107): ESDF, then kVoxels points
    sm = capi.Submap.from_tsdf_layer(ctx, layer, submap_id)
    assert sm.num_blocks() == layer.stats()[0]
    sm.generate_esdf(capi.esdf_config(min_distance_m=0.15, max_distance_m=1.0, default_distance_m=1.0))
    n = sm.extract_voxel_points(1.0, 0.2, True)
    for o in (integ, layer):
        o.destroy()
    return sm, n


def test_scans_to_submaps_to_registration_solve(capi, ctx):
    true = np.array([[0.0, 0.0, 0.0, 0.0], [0.5, 0.3, 0.05, 0.12]])
    traj_a = [(-1.0 + 0.2 * k, -0.5 + 0.05 * k, 0.3, 0.1 * k) for k in range(6)]
    traj_b = [(-0.2 + 0.2 * k, 0.1 - 0.05 * k, 0.35, 0.4 - 0.1 * k) for k in range(6)]
    sm_a, na = _build_submap(capi, ctx, true[0], traj_a, 0)
    sm_b, nb = _build_submap(capi, ctx, true[1], traj_b, 1)
    assert na > 15000 and nb > 15000, (na, nb)
    cfg = capi.default_config(registration_point_type=capi.POINTS_VOXELS)
    cfs = [capi.RegistrationCostFunction(ctx, sm_a, sm_b, cfg),
           capi.RegistrationCostFunction(ctx, sm_b, sm_a, cfg)]
    pairs = [(0, 1), (1, 0)]
    batch = capi.RegistrationBatch(ctx, cfs, pairs)
    backend = GpuBackend(capi, ctx, batch, 2)
    poses0 = true.copy()
    poses0[1] += np.array([0.08, -0.06, 0.04, 0.03])          # odometry drift
    prob = lm.Problem(backend, 2, pairs)
    x, s = lm.solve(prob, poses0, parameter_tolerance=1e-7, function_tolerance=1e-10,
                    max_iterations=40, max_seconds=60)
    err0 = np.abs(poses0[1] - true[1])
    err = np.abs(x[1] - true[1])
    print("drift", err0, "-> after solve", err, s)
    assert s["final_cost"] < 0.5 * s["initial_cost"]
    assert err[:3].max() < 0.3 * VS and err[3] < 0.01         # a third of a voxel, 0.6 deg
    assert err[:3].max() < 0.5 * err0[:3].max()
    for o in [batch] + cfs + [sm_a, sm_b]:
        o.destroy()


def test_lidar_session_through_the_city_config2_miniature(capi, ctx):
    """BASELINE config 2 in miniature (harness/pipeline.py): one lap of a LiDAR drive
    through the analytic city, submaps built scan by scan on the device, finished on the
    device, overlap list from the device, pose graph solved with the fused REG pass."""
    import torch
    from harness import pipeline
    torch.cuda.synchronize()
    out = pipeline.run(capi, ctx, torch, n_submaps=10, scans_per_submap=8, n_az=512, n_el=32, seed=3,
                       isosurface_points=False,      # kVoxels: the steadier mode on this miniature
                       deterministic_tsdf=True)      # the same maps on every run
    print(out)
    assert out["dropped_updates"] == 0
    assert out["voxel_points_per_submap"] > 20000 and out["isosurface_points_per_submap"] > 5000
    # consecutive submaps overlap, and the return leg overlaps the outbound leg (loop closures
    # found by the overlap test alone)
    assert out["overlapping_pairs_final"] >= 10
    assert out["xy_rmse_m_odometry_only"] > 0.15
    assert out["xy_rmse_m_optimised"] < 0.85 * out["xy_rmse_m_odometry_only"]
