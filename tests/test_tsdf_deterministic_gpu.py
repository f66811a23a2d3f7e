"""The reproducible integration mode (vgx_tsdf_config.deterministic).

voxblox's Fast integrator with ONE thread and its fixed ("mixed") visiting order is deterministic --
that is what oracle/tsdf_oracle.c restates, and what finishSubmap() consumes
(voxgraph/src/frontend/submap_collection/voxgraph_submap.cpp:84-107).  In the reproducible mode the
GPU integrates a scan exactly as that single thread does: the same rays are cast (start-voxel
de-duplication through the approximate set, in visiting order), every ray stops at the same voxel
(observed set, > max_consecutive_ray_collisions in a row), every voxel receives the same updates in
the same order, and new blocks take their pool slots in the order a sequential run allocates them.

So the comparison with the oracle is not statistical: block list (order included), distances,
weights and colours are compared BIT FOR BIT on dense scans -- the racing mode can only do that where
the algorithm is order independent (tests/test_tsdf_gpu.py)."""
import numpy as np
import pytest

from oracle import pyoracle as orc

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


def _lidar_scan(n_az, n_el, seed, room=((-5.0, -4.0, -1.0), (5.0, 4.0, 3.0)), origin=(0, 0, 0), el=0.35):
    """rays from `origin` inside an analytic box room, sensor-frame points (sensor axes = room axes)"""
    rng = np.random.default_rng(seed)
    az = np.linspace(-np.pi, np.pi, n_az, endpoint=False) + rng.uniform(0, 1e-3)
    e = np.linspace(-el, el, n_el)
    A, E = np.meshgrid(az, e)
    d = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], -1).reshape(-1, 3)
    lo, hi = np.array(room[0]) - origin, np.array(room[1]) - origin
    with np.errstate(divide="ignore"):
        t = np.where(d > 0, hi / d, lo / d)
    r = t.min(1)
    return (d * r[:, None]).astype(F)


def _rgbd_scan(seed=0):
    """640 x 480 pinhole depth image of a box room (BASELINE config 4 shape), sensor looks along +x"""
    u, v = np.meshgrid((np.arange(640) - 319.5) / 525.0, (np.arange(480) - 239.5) / 525.0)
    d = np.stack([np.ones_like(u), -u, -v], -1).reshape(-1, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    lo, hi = np.array([-5.0, -4.0, -1.0]), np.array([4.0, 4.0, 3.0])
    origin = np.array([-0.5 + 0.01 * seed, 0.3, 0.4])
    with np.errstate(divide="ignore"):
        t = np.where(d > 0, (hi - origin) / d, (lo - origin) / d).min(1)
    return (d * t[:, None]).astype(F), origin.astype(F)


def _assert_layers_identical(ol, gl, what=""):
    obi, od, ow, oc = ol.download()
    gbi, gd, gw, gc = gl.download()
    assert len(obi) == len(gbi), (what, len(obi), len(gbi))
    # same blocks, allocated in the same order (first update in visiting order)
    assert np.array_equal(obi, gbi), (what, "block order", np.flatnonzero((obi != gbi).any(1))[:5])
    for name, a, b in (("distance", od, gd), ("weight", ow, gw)):
        same = a.view(np.uint32) == b.view(np.uint32)
        assert same.all(), (what, name, int((~same).sum()), "of", same.size,
                            a[~same][:4], b[~same][:4])
    assert np.array_equal(oc, gc), (what, "colour", int((oc != gc).any(-1).sum()))
    return len(obi), int((ow > 0).sum())


def test_dense_lidar_scans_are_the_single_thread_result_bit_for_bit(capi, ctx):
    """voxgraph_mapper.yaml:21-28 (0.2 m voxels, 16 m rays, truncation 0.6, constant weight, drop-off,
    sparsity compensation 20), 64 x 1024 returns per scan, eight scans from a moving, turning sensor,
    random colours: the layer after every scan is the oracle's, bit for bit."""
    vs, vps = 0.2, 16
    ocfg = orc.voxgraph_tsdf_config()
    gcfg = capi.voxgraph_tsdf_config(deterministic=1)
    ol, gl = orc.TsdfLayer(vs, vps), capi.TsdfLayer(ctx, vs, vps)
    oi, gi = orc.FastTsdfIntegrator(ocfg, ol), capi.FastTsdfIntegrator(ctx, gcfg, gl)
    rng = np.random.default_rng(7)
    for k in range(8):
        origin = np.array([0.35 * k - 1.0, -0.22 * k + 0.5, 0.03 * k], F)
        pts = _lidar_scan(1024, 64, 10 + k, origin=origin.astype(np.float64))
        yaw = 0.15 * k
        q = np.array([np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)], F)
        # rotate the sensor: points given in the rotated sensor frame
        c, s = np.cos(-yaw), np.sin(-yaw)
        pts = np.stack([c * pts[:, 0] - s * pts[:, 1], s * pts[:, 0] + c * pts[:, 1], pts[:, 2]], 1).astype(F)
        T = np.r_[q, origin].astype(F)
        col = rng.integers(0, 256, (len(pts), 4)).astype(np.uint8)
        a = oi.integratePointCloud(T, pts, col)
        b = gi.integratePointCloud(T, pts, col)
        assert a == b, (k, a, b)
        nb, nobs = _assert_layers_identical(ol, gl, f"scan {k}")
    assert gl.stats()[1] == 0
    print("LiDAR: 8 scans, blocks", nb, "observed voxels", nobs, "updates in the last scan", a)
    assert nobs > 30000
    for o in (gi, gl):
        o.destroy()


def test_the_void_call_returns_with_the_scan_queued_and_its_arrays_consumed(capi, ctx):
    """vgx_tsdf_integrate(n_updates = NULL) -- voxblox's void integratePointCloud -- reads the caller's (pageable) arrays through
    the integrator's two pinned staging buffers and returns with the scan QUEUED: the caller may scribble over the arrays at
    once, call again at once (several scans in flight: the staging buffers alternate), change the scan's size; every reader of
    the layer is ordered behind the scans.  Reproducible mode, so the layer must be the oracle's bit for bit."""
    vs, vps = 0.2, 16
    ol, gl = orc.TsdfLayer(vs, vps), capi.TsdfLayer(ctx, vs, vps)
    oi = orc.FastTsdfIntegrator(orc.voxgraph_tsdf_config(), ol)
    gi = capi.FastTsdfIntegrator(ctx, capi.voxgraph_tsdf_config(deterministic=1), gl)
    rng = np.random.default_rng(11)
    for k in range(7):
        origin = np.array([0.3 * k - 0.8, -0.2 * k + 0.4, 0.02 * k], F)
        pts = _lidar_scan(512 if k % 3 else 768, 32, 40 + k, origin=origin.astype(np.float64))   # (sizes change: buffers regrow)
        T = np.r_[np.array([1, 0, 0, 0], F), origin].astype(F)
        col = rng.integers(0, 256, (len(pts), 4)).astype(np.uint8)
        oi.integratePointCloud(T, pts.copy(), col.copy())
        assert gi.integratePointCloud(T, pts, col, count=False) == 0      # (nothing counted, nothing waited for)
        pts[:] = np.nan                                                    # the arrays are the caller's again
        col[:] = 255
    _assert_layers_identical(ol, gl, "after seven queued scans")
    assert gl.stats()[1] == 0
    for o in (gi, gl):
        o.destroy()


def test_a_session_old_integrator_is_the_oracles_bit_for_bit_and_does_less_work(capi, ctx):
    """The reference keeps ONE FastTsdfIntegrator for the whole mapping session and points it at every new submap's layer
    (pointcloud_integrator.cpp:66-75), and voxblox's ApproxHashSet never forgets between its 10 000-scan resets: the mark
    scan N leaves at slot (hash + N) reads as "present" for the voxel with hash - k in scan N + k, so marks of earlier scans
    cut later scans' rays short (DESIGN.md 3 "The integrator's age"; what bench.py's TSDF figures are measured with).
    Four submaps of ten scans each through one integrator: the reproducible mode's layer and update count are the oracle's
    at every scan of every submap -- the artefact included -- and the same ten scans cost fewer voxel updates in the
    fourth submap than in the first."""
    vs, vps = 0.2, 16
    ocfg = orc.voxgraph_tsdf_config()
    gcfg = capi.voxgraph_tsdf_config(deterministic=1)
    scans = []
    for k in range(10):
        origin = np.array([0.02 * k - 1.0, -0.005 * k + 0.5, 0.001 * k + 0.3], F)
        scans.append((np.r_[np.array([1, 0, 0, 0], F), origin].astype(F), _lidar_scan(512, 32, 40 + k, origin=origin.astype(np.float64))))
    ol, gl = orc.TsdfLayer(vs, vps), capi.TsdfLayer(ctx, vs, vps)
    oi, gi = orc.FastTsdfIntegrator(ocfg, ol), capi.FastTsdfIntegrator(ctx, gcfg, gl)
    updates = []
    old_layers = []
    for submap in range(4):
        if submap:                     # a new submap: a fresh layer, the SAME integrator
            old_layers.append(gl)
            ol, gl = orc.TsdfLayer(vs, vps), capi.TsdfLayer(ctx, vs, vps)
            oi.setLayer(ol)
            gi.setLayer(gl)
        per = []
        for k, (T, pts) in enumerate(scans):
            a = oi.integratePointCloud(T, pts)
            b = gi.integratePointCloud(T, pts)
            assert a == b, (submap, k, a, b)
            per.append(a)
        _assert_layers_identical(ol, gl, f"submap {submap}")
        updates.append(sum(per[1:]))
    print("voxel updates of scans 1..9, submap by submap:", updates)
    assert updates[3] < updates[0], updates      # the same scans, an older integrator: rays stop earlier
    assert gl.stats()[1] == 0
    for o in [gi, gl] + old_layers:
        o.destroy()


def test_rgbd_fullsize_scans_bit_for_bit(capi, ctx):
    """BASELINE config 4: 640 x 480 depth images at 0.05 m voxels, truncation 0.15 m, 5 m rays, 1/z^2
    weights (voxblox's default), three consecutive frames from a slowly moving camera."""
    vs, vps = 0.05, 16
    kw = dict(default_truncation_distance=0.15, max_ray_length_m=5.0)
    ocfg, gcfg = orc.tsdf_config(**kw), capi.tsdf_config(deterministic=1, **kw)
    ol, gl = orc.TsdfLayer(vs, vps), capi.TsdfLayer(ctx, vs, vps)
    oi, gi = orc.FastTsdfIntegrator(ocfg, ol), capi.FastTsdfIntegrator(ctx, gcfg, gl)
    rng = np.random.default_rng(3)
    for k in range(3):
        pts, origin = _rgbd_scan(k)
        assert len(pts) == 307200
        T = np.r_[np.array([1, 0, 0, 0], F), origin].astype(F)
        col = rng.integers(0, 256, (len(pts), 4)).astype(np.uint8)
        a = oi.integratePointCloud(T, pts, col)
        b = gi.integratePointCloud(T, pts, col)
        assert a == b, (k, a, b)
        nb, nobs = _assert_layers_identical(ol, gl, f"frame {k}")
    print("RGB-D: 3 frames, blocks", nb, "observed voxels", nobs, "updates in the last frame", a)
    assert a > 100000 and nobs > 200000
    for o in (gi, gl):
        o.destroy()


def test_five_runs_give_the_same_layer(capi, ctx):
    """Same scans, five runs, on a dense LiDAR and an RGB-D session: every download (block order,
    distance, weight, colour) is byte-identical.  The racing mode, for contrast, differs between runs
    on the same input (reported, not asserted: a race is allowed to come out the same)."""
    import hashlib

    def session(cfg, kind):
        vs = 0.2 if kind == "lidar" else 0.05
        gl = capi.TsdfLayer(ctx, vs, 16)
        gi = capi.FastTsdfIntegrator(ctx, cfg, gl)
        for k in range(3):
            if kind == "lidar":
                origin = np.array([0.4 * k, -0.2 * k, 0.0], F)
                pts = _lidar_scan(1024, 64, 30 + k, origin=origin.astype(np.float64))
            else:
                pts, origin = _rgbd_scan(k)
            gi.integratePointCloud(np.r_[np.array([1, 0, 0, 0], F), origin].astype(F), pts)
        h = hashlib.sha256()
        for a in gl.download():
            h.update(np.ascontiguousarray(a).tobytes())
        for o in (gi, gl):
            o.destroy()
        return h.hexdigest()

    for kind, base in (("lidar", capi.voxgraph_tsdf_config),
                       ("rgbd", lambda **kw: capi.tsdf_config(default_truncation_distance=0.15, **kw))):
        det = {session(base(deterministic=1), kind) for _ in range(5)}
        assert len(det) == 1, (kind, det)
        racing = {session(base(), kind) for _ in range(3)}
        print(kind, "reproducible mode: 5 runs, 1 digest; racing mode:", len(racing), "digest(s) in 3 runs")


ORDERS = {"mixed": (0, 1), "sorted": (1, 2)}     # name -> (vgx_tsdf_config.integration_order, the oracle's)


@pytest.mark.parametrize("order", ["mixed", "sorted"])
def test_randomised_configurations_multi_ray_scans(capi, ctx, order):
    """(both integration_order_mode settings) eight random integrator configurations (vps 8/16, voxel 5-30 cm, carving on/off, constant or 1/z^2
    weights, drop-off, sparsity compensation, low max_weight, max_consecutive_ray_collisions 0-3,
    start-voxel subsampling 1/2/4, allow_clear on/off with returns beyond the maximum range, freespace
    scans, points at the origin / too close / non-finite), four dense scans each."""
    for seed in range(8):
        rng = np.random.default_rng(900 + seed)
        vps = 8 if seed % 2 else 16
        vs = float(rng.choice([0.05, 0.1, 0.2, 0.3]))
        kw = dict(default_truncation_distance=float(rng.uniform(2, 4) * vs),
                  max_ray_length_m=float(rng.uniform(25, 45) * vs),
                  min_ray_length_m=float(rng.uniform(0.5, 2) * vs),
                  voxel_carving_enabled=int(rng.integers(0, 2)), use_const_weight=int(rng.integers(0, 2)),
                  use_weight_dropoff=int(rng.integers(0, 2)),
                  use_sparsity_compensation_factor=int(rng.integers(0, 2)),
                  sparsity_compensation_factor=float(rng.uniform(1, 30)),
                  allow_clear=int(rng.integers(0, 2)), max_weight=float(rng.choice([3.0, 50.0, 10000.0])),
                  max_consecutive_ray_collisions=int(rng.integers(0, 4)),
                  start_voxel_subsampling_factor=float(rng.choice([1.0, 2.0, 4.0])))
        ocfg = orc.tsdf_config(integration_order=ORDERS[order][1], **kw)
        gcfg = capi.tsdf_config(deterministic=1, integration_order=ORDERS[order][0], **kw)
        ol, gl = orc.TsdfLayer(vs, vps), capi.TsdfLayer(ctx, vs, vps)
        oi, gi = orc.FastTsdfIntegrator(ocfg, ol), capi.FastTsdfIntegrator(ctx, gcfg, gl)
        # every other configuration: rays cut after 1-9 steps whatever the scan's size, so that these four scans of
        # equal size go through the extension, the marks kept from scan to scan and the warm second attempt
        if seed % 2:
            gi.set_speculation(int(rng.choice([1, 2, 4, 9])), 0)
        room = ((-30 * vs, -24 * vs, -6 * vs), (32 * vs, 50 * vs, 14 * vs))   # partly beyond max range
        for k in range(4):
            origin = (rng.uniform(-3, 3, 3) * vs).astype(F)
            pts = _lidar_scan(300, 20, seed * 10 + k, room=room, origin=origin.astype(np.float64), el=0.5)
            pts = pts[rng.permutation(len(pts))]                       # not a multiple of 1024: tail in order
            pts[:5] = 0.0                                              # at the sensor: shorter than min_ray
            pts[5:8] *= F(0.3 * kw["min_ray_length_m"]) / np.linalg.norm(pts[5:8], axis=1, keepdims=True).astype(F)
            pts[8] = [np.nan, 1.0, 1.0]
            pts[9] = [np.inf, 0.0, 0.0]
            ang = rng.uniform(-3, 3)
            ax = rng.normal(0, 1, 3); ax /= np.linalg.norm(ax)
            T = np.r_[np.cos(ang / 2), np.sin(ang / 2) * ax, origin].astype(F)
            col = rng.integers(0, 256, (len(pts), 4)).astype(np.uint8)
            free = bool(k == 3)
            a = oi.integratePointCloud(T, pts, col, free)
            b = gi.integratePointCloud(T, pts, col, free)
            assert a == b, (seed, k, a, b, kw)
            _assert_layers_identical(ol, gl, f"seed {seed} scan {k} {kw}")
        assert gl.stats()[1] == 0
        for o in (gi, gl):
            o.destroy()


@pytest.mark.parametrize("depth,threshold", [(1, 0), (2, 0), (3, 0), (6, 0), (12, 0), (32, 0), (4, 20_000), (32, 8 << 20)])
def test_the_layer_does_not_depend_on_how_deep_rays_are_speculated(capi, ctx, depth, threshold):
    """Bounded speculation (det_count_kernel / det_extend_kernel): rays written out `depth` steps deep, extended where
    they ran on; the marks of one scan kept for the next scan of as many points (scans 0-2 here; scan 3 has another
    size and starts without them; scan 4 sees a different scene with stale marks); a second attempt starting from
    the first one's stopping steps.  Whatever the setting, the layer is the oracle's bit for bit after every scan."""
    vs, vps = 0.1, 16
    kw = dict(default_truncation_distance=0.3, max_ray_length_m=6.0, min_ray_length_m=0.1, use_const_weight=0,
              max_consecutive_ray_collisions=2, start_voxel_subsampling_factor=2.0)
    ol, gl = orc.TsdfLayer(vs, vps), capi.TsdfLayer(ctx, vs, vps)
    oi = orc.FastTsdfIntegrator(orc.tsdf_config(**kw), ol)
    gi = capi.FastTsdfIntegrator(ctx, capi.tsdf_config(deterministic=1, **kw), gl)
    gi.set_speculation(depth, threshold)
    rooms = [((-3.0, -2.5, -1.0), (3.5, 2.8, 1.6))] * 4 + [((-1.2, -1.0, -0.6), (1.4, 1.1, 0.9))]
    sizes = [(360, 24)] * 3 + [(301, 17), (301, 17)]
    for k, (room, (n_az, n_el)) in enumerate(zip(rooms, sizes)):
        origin = np.array([0.07 * k, -0.05 * k, 0.02 * k])
        pts = _lidar_scan(n_az, n_el, 40 + k, room=room, origin=origin, el=0.6)
        T = np.r_[np.array([1, 0, 0, 0], F), origin.astype(F)].astype(F)
        col = np.full((len(pts), 4), 200, np.uint8)
        a = oi.integratePointCloud(T, pts, col)
        b = gi.integratePointCloud(T, pts, col)
        assert a == b, (k, a, b)
        _assert_layers_identical(ol, gl, f"speculation {depth}/{threshold}, scan {k}")
    for o in (gi, gl):
        o.destroy()


@pytest.mark.parametrize("merged", [False, True])
@pytest.mark.parametrize("freespace", [False, True])
def test_returns_far_beyond_the_map_are_clipped_not_lost(capi, ctx, merged, freespace):
    """A driver's "no return" codes among the points -- 276 km (beyond the merged integrator's 21-bit voxel keys: such
    points were silently dropped until round 4, found by profiles/fuzz_tsdf.py BIG=1 FIRST=5000), 1e9 m (beyond
    32-bit voxel indices), inf, NaN.  With allow_clear (or in a freespace scan) they are clearing rays clipped to
    max_ray_length_m: both integrators, reproducible mode, equal the oracle bit for bit; getGridIndexFromPoint's
    cast is defined the same way on both sides (NaN -> 0, saturating; oracle/tsdf_oracle.c grid_index)."""
    vs, vps = 0.2, 16
    kw = dict(default_truncation_distance=0.6, max_ray_length_m=6.0, min_ray_length_m=0.2, use_const_weight=1,
              allow_clear=1, enable_anti_grazing=1, max_weight=100.0)
    ol, gl = orc.TsdfLayer(vs, vps), capi.TsdfLayer(ctx, vs, vps)
    oi = orc.FastTsdfIntegrator(orc.tsdf_config(**kw), ol)
    gi = capi.FastTsdfIntegrator(ctx, capi.tsdf_config(deterministic=1, **kw), gl)
    rng = np.random.default_rng(3)
    for k in range(3):
        origin = np.array([0.3 * k, -0.2 * k, 0.1])
        pts = _lidar_scan(256, 12, 70 + k, origin=origin)
        far = rng.choice(len(pts), 40, replace=False)
        unit = pts[far] / np.linalg.norm(pts[far], axis=1, keepdims=True)
        pts[far[:10]] = unit[:10] * F(2.76e5)     # > 2^20 voxels of 0.2 m
        pts[far[10:20]] = unit[10:20] * F(1.0e9)  # > 2^31 voxels
        pts[far[20:25]] = unit[20:25] * F(3.0e30)
        pts[far[25]] = [np.inf, 0.0, 0.0]
        pts[far[26]] = [0.0, -np.inf, 0.0]
        pts[far[27]] = [np.nan, 1.0, 1.0]
        pts[far[28]] = [1.0, np.nan, np.nan]
        pts[far[29:31]] = pts[far[0]]             # the same far voxel three times: one group, its first point
        ang = 0.4 * k
        T = np.r_[np.cos(ang / 2), 0, 0, np.sin(ang / 2), origin].astype(F)
        col = rng.integers(0, 256, (len(pts), 4)).astype(np.uint8)
        if merged:
            a, b = oi.integratePointCloudMerged(T, pts, col, freespace), gi.integratePointCloudMerged(T, pts, col, freespace)
        else:
            a, b = oi.integratePointCloud(T, pts, col, freespace), gi.integratePointCloud(T, pts, col, freespace)
        assert a == b, (k, a, b)
        _assert_layers_identical(ol, gl, f"far returns, merged={merged} freespace={freespace} scan {k}")
    assert gl.stats()[1] == 0
    for o in (gi, gl):
        o.destroy()


def test_layer_grows_under_the_reproducible_mode(capi, ctx):
    """80 single-ray scans while the sensor walks 40 m: the block table is re-boxed and the pool
    enlarged under the reproducible mode as under the racing one, nothing is dropped, and -- single
    rays being order independent -- BOTH modes equal the oracle bit for bit, block order included."""
    vs, vps = 0.2, 16
    kw = dict(default_truncation_distance=0.6, max_ray_length_m=10.0, use_const_weight=1)
    ol = orc.TsdfLayer(vs, vps)
    oi = orc.FastTsdfIntegrator(orc.tsdf_config(**kw), ol)
    gl, gl2 = capi.TsdfLayer(ctx, vs, vps), capi.TsdfLayer(ctx, vs, vps)
    g_det = capi.FastTsdfIntegrator(ctx, capi.tsdf_config(deterministic=1, **kw), gl)
    g_race = capi.FastTsdfIntegrator(ctx, capi.tsdf_config(**kw), gl2)
    rng = np.random.default_rng(11)
    for k in range(80):
        origin = np.array([0.5 * k, 0.1 * k, 0.0], F)
        dirv = rng.normal(0, 1, 3); dirv /= np.linalg.norm(dirv)
        p = (dirv * rng.uniform(0.5, 14.0)).astype(F)[None]
        T = np.r_[np.array([1, 0, 0, 0], F), origin].astype(F)
        col = rng.integers(0, 256, (1, 4)).astype(np.uint8)
        a = oi.integratePointCloud(T, p, col)
        assert a == g_det.integratePointCloud(T, p, col) == g_race.integratePointCloud(T, p, col)
    _assert_layers_identical(ol, gl, "reproducible mode, 40 m walk")
    assert gl.growths() > 0 and gl.stats()[1] == 0
    _assert_layers_identical(ol, gl2, "racing mode, single rays")
    for o in (g_det, g_race, gl, gl2):
        o.destroy()


@pytest.mark.parametrize("const_weight,anti_grazing", [(1, 0), (0, 0), (1, 1)])
def test_merged_integrator_reproducible_mode_bit_for_bit(capi, ctx, const_weight, anti_grazing):
    """voxblox::MergedTsdfIntegrator in the reproducible mode: groups in key order, every voxel takes its
    updates in group order (surface groups, then clearing groups) -- the single thread's order, which is
    the order oracle/tsdf_oracle.c walks.  Dense scans with clearing returns, colours, drop-off, with and
    without anti-grazing: block order, distances, weights and colours equal the oracle's exactly (the
    racing mode reaches 60-97 % identical distances on these scans, tests/test_tsdf_merged_gpu.py)."""
    vs, vps = 0.1, 16
    kw = dict(default_truncation_distance=0.3, max_ray_length_m=6.0, use_const_weight=const_weight,
              use_weight_dropoff=1, enable_anti_grazing=anti_grazing)
    ol, gl = orc.TsdfLayer(vs, vps), capi.TsdfLayer(ctx, vs, vps)
    oi = orc.FastTsdfIntegrator(orc.tsdf_config(**kw), ol)
    gi = capi.FastTsdfIntegrator(ctx, capi.tsdf_config(deterministic=1, **kw), gl)
    rng = np.random.default_rng(21)
    for k in range(3):
        pts = _lidar_scan(360, 48, 40 + k, room=((-4.0, -3.0, -1.0), (4.0, 3.0, 2.0)),
                          origin=np.array([0.3 * k, -0.2 * k, 0.05]))
        pts[::37] *= 4.0                                # returns beyond the maximum range: clearing rays
        pts[5::41] *= 0.01                              # below the minimum range: dropped
        cols = rng.integers(0, 255, (len(pts), 4)).astype(np.uint8)
        T = np.array([np.cos(0.1 * k), 0, 0, np.sin(0.1 * k), 0.3 * k, -0.2 * k, 0.05], F)
        n_o = oi.integratePointCloudMerged(T, pts, cols)
        n_g = gi.integratePointCloudMerged(T, pts, cols)
        assert n_o == n_g > 20000, (k, n_o, n_g)
        nb, nobs = _assert_layers_identical(ol, gl, f"merged scan {k}")
    assert gl.stats()[1] == 0 and nobs > 50000
    for o in (gi, gl):
        o.destroy()


def test_merged_integrator_reproducible_mode_fullsize(capi, ctx):
    """the two BASELINE sensor shapes through the merged integrator, reproducible mode: one 64 x 1024 LiDAR
    sweep with the shipped yaml and one 640 x 480 depth image at 0.05 m, each on top of a previous scan"""
    for kind in ("lidar", "rgbd"):
        if kind == "lidar":
            vs, ocfg, gcfg = 0.2, orc.voxgraph_tsdf_config(), capi.voxgraph_tsdf_config(deterministic=1)
        else:
            vs = 0.05
            kw = dict(default_truncation_distance=0.15, max_ray_length_m=5.0)
            ocfg, gcfg = orc.tsdf_config(**kw), capi.tsdf_config(deterministic=1, **kw)
        ol, gl = orc.TsdfLayer(vs, 16), capi.TsdfLayer(ctx, vs, 16)
        oi, gi = orc.FastTsdfIntegrator(ocfg, ol), capi.FastTsdfIntegrator(ctx, gcfg, gl)
        for k in range(2):
            if kind == "lidar":
                origin = np.array([0.4 * k, -0.2 * k, 0.0], F)
                pts = _lidar_scan(1024, 64, 60 + k, origin=origin.astype(np.float64))
            else:
                pts, origin = _rgbd_scan(k)
            T = np.r_[np.array([1, 0, 0, 0], F), origin].astype(F)
            a = oi.integratePointCloudMerged(T, pts)
            b = gi.integratePointCloudMerged(T, pts)
            assert a == b > 0, (kind, k, a, b)
            nb, nobs = _assert_layers_identical(ol, gl, f"merged {kind} scan {k}")
        print("merged", kind, "updates per scan", a, "blocks", nb, "observed voxels", nobs)
        for o in (gi, gl):
            o.destroy()


@pytest.mark.parametrize("order", ["mixed", "sorted"])
def test_small_scans_and_sets_that_are_not_reset_every_frame(capi, ctx, order):
    """(both visiting orders) clear_checks_every_n_frames = 3: the approximate sets keep their contents (and their offset) over
    three scans, so what a ray finds in a slot may have been written by an EARLIER scan -- the
    reproducible mode reads the sets' state where its own scan has no predecessor in a slot and leaves them
    as the single thread would.  Scans of 0, 1, 2, 5, 1023, 1024, 1025 and 3000 points (the mixed
    visiting order groups points by 1024) while the sensor moves: bit for bit after every scan."""
    vs, vps = 0.1, 16
    kw = dict(default_truncation_distance=0.3, max_ray_length_m=8.0, use_const_weight=1,
              clear_checks_every_n_frames=3, max_consecutive_ray_collisions=1)
    ol, gl = orc.TsdfLayer(vs, vps), capi.TsdfLayer(ctx, vs, vps)
    oi = orc.FastTsdfIntegrator(orc.tsdf_config(integration_order=ORDERS[order][1], **kw), ol)
    gi = capi.FastTsdfIntegrator(ctx, capi.tsdf_config(deterministic=1, integration_order=ORDERS[order][0], **kw), gl)
    rng = np.random.default_rng(5)
    full = _lidar_scan(300, 10, 77, room=((-3.0, -2.5, -1.0), (3.5, 2.0, 1.5)), el=0.4)
    for k, n in enumerate([0, 1, 2, 5, 1023, 1024, 1025, 3000, 3000, 3000, 1, 3000]):
        origin = np.array([0.02 * k, -0.015 * k, 0.01 * k], F)
        pts = full[rng.permutation(len(full))[:n]] - origin
        T = np.r_[np.array([1, 0, 0, 0], F), origin].astype(F)
        col = rng.integers(0, 256, (n, 4)).astype(np.uint8)
        a = oi.integratePointCloud(T, pts.reshape(-1, 3), col if n else None)
        b = gi.integratePointCloud(T, pts.reshape(-1, 3), col if n else None)
        assert a == b, (k, n, a, b)
        if ol.num_blocks():
            _assert_layers_identical(ol, gl, f"scan {k} of {n} points")
        else:
            assert gl.stats()[0] == 0
    assert gl.stats()[1] == 0 and ol.num_blocks() > 20
    for o in (gi, gl):
        o.destroy()


@pytest.mark.parametrize("kind", ["lidar", "rgbd"])
@pytest.mark.parametrize("merged", [False, True])
def test_sorted_integration_order_bit_for_bit(capi, ctx, kind, merged):
    """integration_order_mode "sorted" (voxgraph/config/voxgraph_mapper.yaml:29): points visited
    by ascending f32 squaredNorm of point_C, equal ranges by ascending index.  The two BASELINE sensor shapes at
    full size -- a 64 x 1024 LiDAR sweep with the shipped yaml (plus 3000 duplicated returns with other colours:
    exact ties) and a 640 x 480 depth image at 0.05 m -- two scans each, fast and merged integrators, in BOTH
    orders: block list, distances, weights and colours equal the oracle's bit for bit after every scan, and the
    two orders give different layers (the option is not ignored)."""
    import hashlib
    digests = {}
    for order in ("mixed", "sorted"):
        if kind == "lidar":
            vs = 0.2
            ocfg = orc.voxgraph_tsdf_config(integration_order=ORDERS[order][1])
            gcfg = capi.voxgraph_tsdf_config(deterministic=1, integration_order=ORDERS[order][0])
        else:
            vs = 0.05
            kw = dict(default_truncation_distance=0.15, max_ray_length_m=5.0)
            ocfg = orc.tsdf_config(integration_order=ORDERS[order][1], **kw)
            gcfg = capi.tsdf_config(deterministic=1, integration_order=ORDERS[order][0], **kw)
        ol, gl = orc.TsdfLayer(vs, 16), capi.TsdfLayer(ctx, vs, 16)
        oi, gi = orc.FastTsdfIntegrator(ocfg, ol), capi.FastTsdfIntegrator(ctx, gcfg, gl)
        rng = np.random.default_rng(31)
        for k in range(2):
            if kind == "lidar":
                origin = np.array([0.4 * k, -0.2 * k, 0.0], F)
                pts = _lidar_scan(1024, 64, 80 + k, origin=origin.astype(np.float64))
                pts = np.concatenate([pts, pts[rng.integers(0, len(pts), 3000)]])      # exact ties in range
            else:
                pts, origin = _rgbd_scan(k)
            T = np.r_[np.array([1, 0, 0, 0], F), origin].astype(F)
            col = rng.integers(0, 256, (len(pts), 4)).astype(np.uint8)
            a = (oi.integratePointCloudMerged if merged else oi.integratePointCloud)(T, pts, col)
            b = (gi.integratePointCloudMerged if merged else gi.integratePointCloud)(T, pts, col)
            assert a == b > 0, (kind, merged, order, k, a, b)
            _assert_layers_identical(ol, gl, f"{kind} merged={merged} order={order} scan {k}")
        assert gl.stats()[1] == 0
        h = hashlib.sha256()
        for arr in gl.download():
            h.update(np.ascontiguousarray(arr).tobytes())
        digests[order] = h.hexdigest()
        for o in (gi, gl):
            o.destroy()
    assert digests["mixed"] != digests["sorted"], (kind, merged)


def test_racing_mode_ignores_the_integration_order_and_bad_values_are_refused(capi, ctx):
    """the racing fast integrator has no visiting order: integration_order sorted changes nothing it can be held
    to (single-ray scans are order independent: equal to the oracle either way); a value that is neither mixed
    nor sorted is refused by the modes that use it"""
    vs = 0.2
    pts = _lidar_scan(64, 4, 3)
    T = np.array([1, 0, 0, 0, 0, 0, 0], F)
    for order in (0, 1):
        ol, gl = orc.TsdfLayer(vs, 16), capi.TsdfLayer(ctx, vs, 16)
        oi = orc.FastTsdfIntegrator(orc.voxgraph_tsdf_config(), ol)
        gi = capi.FastTsdfIntegrator(ctx, capi.voxgraph_tsdf_config(integration_order=order), gl)
        for p in pts[:40]:
            assert oi.integratePointCloud(T, p[None]) == gi.integratePointCloud(T, p[None])
        _assert_layers_identical(ol, gl, f"racing mode, integration_order {order}")
        for o in (gi, gl):
            o.destroy()
    gl = capi.TsdfLayer(ctx, vs, 16)
    for det, merged in ((1, False), (0, True)):
        gi = capi.FastTsdfIntegrator(ctx, capi.voxgraph_tsdf_config(deterministic=det, integration_order=7), gl)
        with pytest.raises(capi.VgxError) as e:
            (gi.integratePointCloudMerged if merged else gi.integratePointCloud)(T, pts)
        assert e.value.code == -1 and "integration_order" in str(e.value)        # VGX_ERR_INVALID
        gi.destroy()
    gl.destroy()


def test_reproducible_mode_matches_the_committed_oracle_digests(capi, ctx, golden_dir):
    """the same three sessions as tests/test_oracle_tsdf.py's golden test, on the device in the
    reproducible mode: update counts, block count and the SHA-256 of the downloaded layer (block list,
    distances, weights, colours) equal the committed digests of the oracle"""
    import json
    import os
    from tests.golden import make_tsdf_golden as G
    want = json.load(open(os.path.join(golden_dir, "tsdf_oracle_digests.json")))
    for name, kw, merged, scans in G.sessions():
        made = []

        def make_layer(vs, vps):
            made.append(capi.TsdfLayer(ctx, vs, vps))
            return made[-1]

        def make_integ(kw_, layer):
            made.append(capi.FastTsdfIntegrator(ctx, capi.tsdf_config(deterministic=1, **kw_), layer))
            return made[-1]
        got = G.run(make_layer, make_integ, kw, merged, scans)
        assert got == want[name], (name, got, want[name])
        for o in reversed(made):
            o.destroy()
