"""The racing TSDF kernel -- the DEFAULT mode, what voxgraph would run -- held to the sequential integrator BIT FOR BIT on
dense scans with the shipped early-out: trace and replay.

voxblox::FastTsdfIntegrator::integratePointCloud run by worker threads (the call of
voxgraph/src/frontend/measurement_processors/pointcloud_integrator.cpp:66-83) has no single result, so a racing kernel
cannot be compared with ONE oracle layer; rounds 1-5 compared it exactly where order cannot matter and statistically
elsewhere.  Here every scan is run by the event-logging instantiation of the SHIPPED kernel template
(csrc/vgx_tsdf_coop_kernel.h, TRACE = true; include/voxgraph_amd_bench.h) and its log is replayed through the oracle's own
per-point / per-voxel functions (oracle/tsdf_replay.c): per approximate-set slot the exchanges form one path from the
content before the scan to the content after it; every ray's cast / step / stop decisions are the oracle's function of
the values IT got; per voxel the folds form one path from the word before to the word after and each equals the oracle's
updateTsdfVoxel chain over its records; the union of the folds' records is every update every ray must emit, none
twice.  The kernel's two stated liberties are COUNTED in the report (start-set skips next to a lane with the same value;
exchanges behind a stop), never hidden.  tests/test_tsdf_replay_cpu.py shows the checker rejects a log with one event
dropped, duplicated or altered."""
import numpy as np
import pytest

from harness.bench_tsdf import sensor_cases, session_scans
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu
F = np.float32
TRACE_WORDS = 24 << 20     # 192 MB of log: a depth image's ~10 M words with room to spare


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


def _replayed_scan(capi, ocfg, vs, vps, layer, integ, T, pts, col=None, freespace=False, d_points=None):
    """one traced scan: state before, scan, log, state after -> the checker's report (asserted legal)"""
    start0, obs0, _ = integ.download_sets()
    layer0 = layer.download()
    if d_points is None:
        updates = integ.integratePointCloud(T, pts, col, freespace)           # counted: the STATS form of the kernel
    else:
        integ.integrate_device(T, d_points, None, len(pts), freespace)       # uncounted: the form bench.py times
        updates = None
    trace, lost = integ.read_event_trace()
    assert lost == 0, lost
    start1, obs1, (off_s, off_o, _) = integ.download_sets()
    assert layer.stats()[1] == 0                                              # nothing dropped
    rep = orc.tsdf_replay_check(ocfg, vs, vps, T, pts, col, freespace, (off_s, off_o), (start0, obs0), (start1, obs1),
                                layer0, layer.download(), trace)
    assert rep["errors"] == 0, rep["first_error"]
    assert rep["required_updates"] == rep["fold_records"]
    if updates is not None:
        assert rep["required_updates"] == updates, (rep["required_updates"], updates)
    return rep


def _summary(name, reps):
    keys = ("valid_points", "start_skips", "rays_cast", "observed_exchanges", "overrun_exchanges", "rays_with_overrun",
            "max_overrun", "required_updates", "fold_events", "folds_left_alone", "longest_fold", "voxels_with_several_links",
            "colour_writes", "new_blocks")
    tot = {k: (max if k in ("max_overrun", "longest_fold") else sum)(r[k] for r in reps) for k in keys}
    print(f"{name}: {len(reps)} scans replayed, 0 violations;", ", ".join(f"{k} {v}" for k, v in tot.items()))
    return tot


def test_lidar_session_with_the_shipped_yaml_is_a_legal_interleaving(capi, ctx):
    """BASELINE config 2's sensor shape (64 x 1024, voxgraph_mapper.yaml:21-28): a fresh integrator's first scans (long
    walks, every block new), then -- after 90 untraced scans: the regime the reference runs in -- an old one's"""
    dirs, vs, kw, lut_min, lut_dim = sensor_cases()["lidar_64x1024_0.20m_voxgraph_yaml"]
    poses, clouds = session_scans(dirs, 8)
    ocfg, gcfg = orc.tsdf_config(**kw), capi.tsdf_config(**kw)
    layer = capi.TsdfLayer(ctx, vs, 16, lut_min, lut_dim, 256)
    integ = capi.FastTsdfIntegrator(ctx, gcfg, layer)
    integ.set_event_trace(TRACE_WORDS)
    rng = np.random.default_rng(3)
    reps = []
    for k, (T, pts) in enumerate(zip(poses, clouds)):
        col = rng.integers(0, 256, (len(pts), 4)).astype(np.uint8) if k % 2 else None
        reps.append(_replayed_scan(capi, ocfg, vs, 16, layer, integ, T, pts, col))
    fresh = _summary("lidar, fresh integrator", reps)
    assert fresh["rays_cast"] > 8 * 2000 and fresh["voxels_with_several_links"] > 1000 and fresh["new_blocks"] > 20
    integ.set_event_trace(0)
    for _ in range(11):
        for T, pts in zip(poses, clouds):
            integ.integratePointCloud(T, pts)
    integ.set_event_trace(TRACE_WORDS)
    reps = [_replayed_scan(capi, ocfg, vs, 16, layer, integ, T, pts) for T, pts in zip(poses[:4], clouds[:4])]
    old = _summary("lidar, integrator 96 scans old", reps)
    assert old["rays_cast"] > 4 * 1000
    for o in (integ, layer):
        o.destroy()


def test_depth_images_are_a_legal_interleaving(capi, ctx):
    """BASELINE config 4's sensor shape (640 x 480 at 0.05 m voxels, 1/z^2 weights, colours): the scan with the most
    contended voxels (every ray ends next to the sensor) and the longest folds"""
    dirs, vs, kw, lut_min, lut_dim = sensor_cases()["rgbd_640x480_0.05m"]
    poses, clouds = session_scans(dirs, 3)
    ocfg, gcfg = orc.tsdf_config(**kw), capi.tsdf_config(**kw)
    layer = capi.TsdfLayer(ctx, vs, 16, lut_min, lut_dim, 2048)
    integ = capi.FastTsdfIntegrator(ctx, gcfg, layer)
    integ.set_event_trace(TRACE_WORDS)
    rng = np.random.default_rng(4)
    reps = []
    for k, (T, pts) in enumerate(zip(poses, clouds)):
        col = rng.integers(0, 256, (len(pts), 4)).astype(np.uint8)
        reps.append(_replayed_scan(capi, ocfg, vs, 16, layer, integ, T, pts, col))
    tot = _summary("depth image", reps)
    assert tot["rays_cast"] > 3 * 20000 and tot["colour_writes"] > 10000
    for o in (integ, layer):
        o.destroy()


@pytest.mark.parametrize("width", [1024, 1000])
def test_organised_clouds_are_a_legal_interleaving(capi, ctx, width):
    """vgx_tsdf_integrator_set_cloud_width: 16 x 16 tiles of beams per workgroup -- 1024 = whole tiles, 1000 = ragged tiles at
    the right edge (lanes without a point next to lanes with one: the start-set shuffle of ADVICE r5)"""
    dirs, vs, kw, lut_min, lut_dim = sensor_cases()["lidar_64x1024_0.20m_voxgraph_yaml"]
    rows = 64 if width == 1024 else 60
    poses, clouds = session_scans(dirs, 5)
    clouds = [c.reshape(64, 1024, 3)[:rows, :width].reshape(-1, 3).copy() for c in clouds]
    ocfg, gcfg = orc.tsdf_config(**kw), capi.tsdf_config(**kw)
    layer = capi.TsdfLayer(ctx, vs, 16, lut_min, lut_dim, 256)
    integ = capi.FastTsdfIntegrator(ctx, gcfg, layer)
    integ.set_cloud_width(width)
    integ.set_event_trace(TRACE_WORDS)
    reps = [_replayed_scan(capi, ocfg, vs, 16, layer, integ, T, pts) for T, pts in zip(poses, clouds)]
    tot = _summary(f"organised cloud, width {width}", reps)
    assert tot["rays_cast"] > 5 * 1000 and tot["start_skips"] > 1000
    for o in (integ, layer):
        o.destroy()


def test_city_scans_handed_on_between_passes_are_a_legal_interleaving(capi, ctx):
    """BASELINE config 2's stand-in session (harness/pipeline.py: 16 m rays down city streets, 60 % of the points cast a
    ray, more than 128 rays per workgroup): the passes with fewer than eight lanes per ray and the hand-on of rays still
    walking, through the UNCOUNTED kernel and a device pointer -- what bench.py times"""
    import torch
    from harness.pipeline import session_sensor_poses, _inv_compose
    vs, n_az, n_el = 0.2, 1024, 64
    ocfg, gcfg = orc.voxgraph_tsdf_config(), capi.voxgraph_tsdf_config()
    sensor = session_sensor_poses(30, 100)
    P = sensor[0].copy()
    P[2] = 0.0
    layer = capi.TsdfLayer(ctx, vs, 16)
    integ = capi.FastTsdfIntegrator(ctx, gcfg, layer)
    integ.set_event_trace(TRACE_WORDS)
    d_pts = torch.empty((n_az * n_el, 3), dtype=torch.float32, device="cuda")
    reps = []
    for j in (0, 1, 2, 40, 41):
        capi.synth_city_scan(ctx, sensor[j], n_az, n_el, np.deg2rad(33.2), 40.0, 2, d_pts.data_ptr())
        ctx.synchronize()
        pts = d_pts.cpu().numpy()
        T = np.asarray(_inv_compose(P, sensor[j]), F)
        reps.append(_replayed_scan(capi, ocfg, vs, 16, layer, integ, T, pts, d_points=d_pts.data_ptr()))
    tot = _summary("city scans (config-2 stand-in)", reps)
    assert tot["rays_cast"] > 5 * 20000 and tot["longest_fold"] >= 2
    for o in (integ, layer):
        o.destroy()


def test_random_integrator_configurations_are_legal_interleavings(capi, ctx):
    """the options that change what a ray does: collision limits 0-3, carving on / off, start-voxel subsampling, weight
    drop-off, sparsity compensation, low max_weight, clearing and free-space scans, degenerate points, vps 8 / 16,
    clear_checks_every_n_frames > 1 (a scan that does NOT reset the sets)"""
    rng = np.random.default_rng(17)
    az = np.linspace(-np.pi, np.pi, 256, endpoint=False) + 0.003
    el = np.linspace(-0.5, 0.5, 24)
    A, E = np.meshgrid(az, el)
    d = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], -1).reshape(-1, 3)
    total = 0
    for case in range(10):
        vps = int(rng.choice([8, 16]))
        vs = float(rng.choice([0.1, 0.2]))
        kw = dict(default_truncation_distance=float(rng.choice([2, 3])) * vs, max_ray_length_m=float(rng.choice([4.0, 8.0])),
                  voxel_carving_enabled=int(rng.integers(0, 2)), use_const_weight=int(rng.integers(0, 2)),
                  use_weight_dropoff=int(rng.integers(0, 2)), use_sparsity_compensation_factor=int(rng.integers(0, 2)),
                  sparsity_compensation_factor=float(rng.choice([1.0, 20.0])), allow_clear=int(rng.integers(0, 2)),
                  start_voxel_subsampling_factor=float(rng.choice([1.0, 2.0, 4.0])),
                  max_consecutive_ray_collisions=int(rng.integers(0, 4)), max_weight=float(rng.choice([3.0, 10000.0])),
                  clear_checks_every_n_frames=int(rng.choice([1, 1, 3])))
        ocfg, gcfg = orc.tsdf_config(**kw), capi.tsdf_config(**kw)
        layer = capi.TsdfLayer(ctx, vs, vps)
        integ = capi.FastTsdfIntegrator(ctx, gcfg, layer)
        integ.set_event_trace(TRACE_WORDS)
        reps = []
        for k in range(4):
            origin = np.array([0.2 * k, -0.1 * k, 0.03 * k])
            lo, hi = np.array([-5.0, -4.0, -1.0]) - origin, np.array([5.0, 4.0, 3.0]) - origin
            t = np.where(d > 0, hi / d, lo / d).min(1)
            pts = (d * t[:, None]).astype(F)
            pts[rng.integers(0, len(pts), 30)] *= F(3.0)                      # beyond max_ray_length: clearing / dropped
            pts[rng.integers(0, len(pts), 10)] = 0                            # degenerate
            pts[rng.integers(0, len(pts), 3)] = np.nan
            col = rng.integers(0, 256, (len(pts), 4)).astype(np.uint8)
            yaw = 0.1 * k
            T = np.array([np.cos(yaw / 2), 0, 0, np.sin(yaw / 2), *origin], F)
            reps.append(_replayed_scan(capi, ocfg, vs, vps, layer, integ, T, pts, col, freespace=bool(k == 3 and case % 2)))
        total += sum(r["required_updates"] for r in reps)
        for o in (integ, layer):
            o.destroy()
    print("random configurations: 40 scans replayed,", total, "updates, 0 violations")
    assert total > 100000
