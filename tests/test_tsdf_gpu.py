"""GPU parity tests of the TSDF path (voxblox::FastTsdfIntegrator semantics).

The reference integrator is nondeterministic by construction (worker threads racing
on approximate hash sets and per-voxel locks), so parity is layered:
  * EXACT (bit-for-bit vs the single-thread oracle) wherever the algorithm is
    order-independent: sequences of single-ray scans, and multi-ray scans whose
    rays share no voxel;
  * order-independent invariants on dense scans with the early-out disabled;
  * statistical agreement on a full LiDAR-like scan with the shipped yaml."""
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


def _both_cfg(capi, **kw):
    o = orc.tsdf_config(**kw)
    g = capi.tsdf_config(**kw)
    return o, g


def _as_dict(bi, d, w, rgba, vps):
    out = {}
    for b in range(len(bi)):
        base = bi[b].astype(np.int64) * vps
        lin = np.arange(vps ** 3)
        keys = np.stack([base[0] + lin % vps, base[1] + (lin // vps) % vps, base[2] + lin // (vps * vps)], 1)
        for k, dd, ww, cc in zip(map(tuple, keys), d[b], w[b], rgba[b]):
            out[k] = (dd, ww, tuple(cc))
    return out


def _grids(bi, d, w, vps, lo, hi):
    """dense [X,Y,Z] arrays over voxel box [lo,hi) from a block list"""
    shape = tuple(np.array(hi) - np.array(lo))
    D = np.zeros(shape, F)
    W = np.zeros(shape, F)
    A = np.zeros(shape, bool)
    for b in range(len(bi)):
        o = bi[b].astype(np.int64) * vps - np.array(lo)
        blkd = d[b].reshape(vps, vps, vps).transpose(2, 1, 0)      # [x,y,z]
        blkw = w[b].reshape(vps, vps, vps).transpose(2, 1, 0)
        sl = tuple(slice(o[a], o[a] + vps) for a in range(3))
        D[sl], W[sl], A[sl] = blkd, blkw, True
    return D, W, A


def _rot_z(yaw):
    return np.array([np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)], F)


def test_sequence_of_single_ray_scans_is_bit_exact(capi, ctx):
    """n = 1 per scan removes every race: the GPU must reproduce the oracle bit for bit,
    including the approximate-set offset artefact between consecutive scans, weight
    drop-off, sparsity compensation, clearing rays and colour blending."""
    vs, vps = 0.2, 16
    ocfg, gcfg = _both_cfg(capi, default_truncation_distance=0.6, max_ray_length_m=16.0,
                           use_const_weight=1, use_weight_dropoff=1,
                           use_sparsity_compensation_factor=1, sparsity_compensation_factor=20.0)
    ol = orc.TsdfLayer(vs, vps)
    oi = orc.FastTsdfIntegrator(ocfg, ol)
    gl = capi.TsdfLayer(ctx, vs, vps, (-8, -8, -8), (16, 16, 16), 1024)
    gi = capi.FastTsdfIntegrator(ctx, gcfg, gl)
    rng = np.random.default_rng(0)
    for k in range(120):
        yaw = rng.uniform(-3, 3)
        T = np.concatenate([_rot_z(yaw), rng.uniform(-1, 1, 3)]).astype(F)
        r = rng.uniform(0.05, 20.0)                       # includes too-short and clearing rays
        dirv = rng.normal(0, 1, 3); dirv /= np.linalg.norm(dirv)
        p = (dirv * r).astype(F)[None]
        col = rng.integers(0, 256, (1, 4)).astype(np.uint8)
        a = oi.integratePointCloud(T, p, col)
        b = gi.integratePointCloud(T, p, col)
        assert a == b, (k, a, b)
    obi, od, ow, oc = ol.download()
    gbi, gd, gw, gc = gl.download()
    assert gl.stats()[1] == 0
    assert set(map(tuple, obi)) == set(map(tuple, gbi))
    O, G = _as_dict(obi, od, ow, oc, vps), _as_dict(gbi, gd, gw, gc, vps)
    bad = [k for k in O if not (O[k][0] == G[k][0] and O[k][1] == G[k][1] and O[k][2] == G[k][2])]
    assert not bad, (len(bad), bad[:3], [O[k] for k in bad[:3]], [G[k] for k in bad[:3]])
    for o in (gi, gl):
        o.destroy()


def test_disjoint_rays_without_carving_are_bit_exact(capi, ctx):
    vs, vps = 0.1, 16
    kw = dict(default_truncation_distance=0.25, voxel_carving_enabled=0, use_const_weight=0,
              max_ray_length_m=30.0)
    ocfg, gcfg = _both_cfg(capi, **kw)
    # points on a coarse lattice (1.3 m apart): truncation bands never meet
    g = np.arange(-4, 5) * 1.3
    pts = np.array([(x + 0.03, y - 0.02, 6.0 + 0.1 * np.sin(x * y)) for x in g for y in g], F)
    T = np.array([1, 0, 0, 0, 0.07, -0.03, 0.02], F)
    ol = orc.TsdfLayer(vs, vps)
    a = orc.FastTsdfIntegrator(ocfg, ol).integratePointCloud(T, pts)
    gl = capi.TsdfLayer(ctx, vs, vps, (-8, -8, -2), (16, 16, 12), 2048)
    gi = capi.FastTsdfIntegrator(ctx, gcfg, gl)
    b = gi.integratePointCloud(T, pts)
    assert a == b and a > 300
    O = _as_dict(*ol.download(), vps)
    G = _as_dict(*gl.download(), vps)
    assert O.keys() == G.keys()
    assert all(O[k][0] == G[k][0] and O[k][1] == G[k][1] for k in O)
    for o in (gi, gl):
        o.destroy()


def _lidar_scan(n_az, n_el, seed):
    """rays from inside a 10 x 8 x 4 m room (analytic box), sensor-frame points"""
    rng = np.random.default_rng(seed)
    az = np.linspace(-np.pi, np.pi, n_az, endpoint=False) + rng.uniform(0, 1e-3)
    el = np.linspace(-0.35, 0.35, n_el)
    A, E = np.meshgrid(az, el)
    d = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], -1).reshape(-1, 3)
    lo, hi = np.array([-5.0, -4.0, -1.0]), np.array([5.0, 4.0, 3.0])
    with np.errstate(divide="ignore"):
        t = np.where(d > 0, hi / d, lo / d)
    r = t.min(1)
    return (d * r[:, None]).astype(F)


@pytest.mark.parametrize("cloud_width", [0, 40])
def test_dense_scan_invariants_with_early_out_disabled(capi, ctx, cloud_width):
    """max_consecutive_ray_collisions = huge, constant weight, no drop-off / sparsity:
    every ray updates every voxel it crosses, so weight(v) = #rays through v exactly
    (integer sums: order-independent) and free space is exactly +truncation.
    cloud_width != 0: the same scan declared an organised cloud of that many points per row
    (vgx_tsdf_integrator_set_cloud_width: 16 x 16 tiles of beams per workgroup, here with ragged tiles at the
    right and bottom edges) -- only the assignment of points to workgroups may change, none of the facts below."""
    vs, vps, trunc = 0.1, 16, 0.3
    kw = dict(default_truncation_distance=trunc, max_ray_length_m=20.0, use_const_weight=1,
              use_weight_dropoff=0, max_consecutive_ray_collisions=1 << 30)
    ocfg, gcfg = _both_cfg(capi, **kw)
    pts = _lidar_scan(360, 24, 1)
    # keep one point per start cell so the start-voxel dedup keeps the same rays
    cells = np.floor(pts * np.float32(2.0 / vs) + 1e-6).astype(np.int64)
    _, first = np.unique(cells, axis=0, return_index=True)
    pts = pts[np.sort(first)]
    if cloud_width:
        # whole rows: padded with points the integrator skips (shorter than min_ray_length_m)
        pad = (-len(pts)) % cloud_width
        pts = np.concatenate([pts, np.zeros((pad, 3), F)])
    T = np.array([1, 0, 0, 0, 0, 0, 0], F)
    ol = orc.TsdfLayer(vs, vps)
    a = orc.FastTsdfIntegrator(ocfg, ol).integratePointCloud(T, pts)
    gl = capi.TsdfLayer(ctx, vs, vps, (-5, -4, -2), (10, 8, 5), 400)
    gi = capi.FastTsdfIntegrator(ctx, gcfg, gl)
    gi.set_cloud_width(cloud_width)
    b = gi.integratePointCloud(T, pts)
    assert gl.stats()[1] == 0
    assert a == b                                   # same number of voxel updates
    lo, hi = (-80, -64, -32), (80, 64, 48)
    oD, oW, oA = _grids(*ol.download()[:3], vps, lo, hi)
    gD, gW, gA = _grids(*gl.download()[:3], vps, lo, hi)
    assert np.array_equal(oA, gA)
    assert np.array_equal(oW, gW)                   # exact: integer ray counts
    # a voxel crossed by exactly ONE ray took one update: no order to depend on -- bit for bit (an
    # exactness check the racing mode can be held to on a dense scan)
    one = oW == 1
    assert one.sum() > 1000 and np.array_equal(oD[one].view(np.uint32), gD[one].view(np.uint32)), one.sum()
    diff = np.abs(gD - oD)[oW > 0]
    print("dense scan: updates", a, b, "voxels", (oW > 0).sum(), "p99/p99.9/max |dd|",
          np.percentile(diff, 99), np.percentile(diff, 99.9), diff.max())
    # voxels deeper than truncation + a voxel diagonal inside the room only ever see
    # sdf > truncation: exactly +truncation whatever the order
    c = [(np.arange(lo[k], hi[k]) + 0.5) * vs for k in range(3)]
    X, Y, Z = np.meshgrid(*c, indexing="ij")
    wall = np.minimum.reduce([5 - np.abs(X), 4 - np.abs(Y), Z + 1, 3 - Z])
    free = (oW > 0) & (wall > trunc + 2 * vs)
    assert free.sum() > 10000 and np.all(oD[free] == F(trunc)) and np.all(gD[free] == F(trunc))
    # elsewhere the running average is clamped after every update, so the order of
    # updates matters at the edge of the truncation band (and only there)
    assert np.percentile(diff, 99) < 1e-4 and diff.max() <= 2 * trunc
    for o in (gi, gl):
        o.destroy()


def test_full_scan_with_shipped_config_agrees_statistically(capi, ctx):
    """voxgraph_mapper.yaml:21-28 on a LiDAR-like scan, several scans from moving poses:
    the GPU (65k concurrent rays) and the single-thread oracle are two interleavings of
    the same algorithm."""
    vs, vps, trunc = 0.2, 16, 0.6
    ocfg, gcfg = orc.voxgraph_tsdf_config(), capi.voxgraph_tsdf_config()
    ocfg_seq = orc.voxgraph_tsdf_config(integration_order=0)
    ol, ol2 = orc.TsdfLayer(vs, vps), orc.TsdfLayer(vs, vps)
    oi, oi2 = orc.FastTsdfIntegrator(ocfg, ol), orc.FastTsdfIntegrator(ocfg_seq, ol2)
    scans = []
    tot_o = tot_o2 = 0
    for k in range(5):
        pts = _lidar_scan(1024, 64, 10 + k)
        T = np.array([1, 0, 0, 0, 0.15 * k, -0.1 * k, 0.02 * k], F)
        pts = (pts - T[4:]).astype(F)               # same room seen from the moved sensor
        scans.append((T, pts))
        tot_o += oi.integratePointCloud(T, pts)
        tot_o2 += oi2.integratePointCloud(T, pts[::-1].copy())   # another legal order
    lo, hi = (-48, -48, -32), (48, 48, 32)
    oD, oW, oA = _grids(*ol.download()[:3], vps, lo, hi)
    pD, pW, pA = _grids(*ol2.download()[:3], vps, lo, hi)
    ref_both = (oW > 0) & (pW > 0)
    q = [50, 90, 99]
    # THREE racing sessions of the same scans (a race comes out differently every time): every one must pass the
    # structural checks, and the MEDIAN of their error percentiles the tight bound (the bound was
    # loosened to 2 x + 2 % in round 3 to keep single runs green; the median of three holds 1.5 x + 1 %)
    runs = []
    for run in range(3):
        gl = capi.TsdfLayer(ctx, vs, vps, (-3, -3, -2), (6, 6, 4), 144)
        gi = capi.FastTsdfIntegrator(ctx, gcfg, gl)
        tot_g = sum(gi.integratePointCloud(T, pts) for T, pts in scans)
        assert gl.stats()[1] == 0
        gD, gW, gA = _grids(*gl.download()[:3], vps, lo, hi)
        print("run", run, "updates oracle/oracle2/gpu:", tot_o, tot_o2, tot_g, "blocks:", oA.sum() // 4096, gA.sum() // 4096)
        assert np.array_equal(oA, gA)                               # same blocks allocated
        assert abs(tot_g - tot_o) < max(0.25 * tot_o, 2 * abs(tot_o2 - tot_o))
        both = (oW > 0) & (gW > 0)
        # which free-space voxels a ray still reaches before its early-out depends on the
        # interleaving; the observed regions overlap, they are not identical
        assert both.sum() > 0.90 * ref_both.sum()
        band = both & ref_both & (np.abs(oD) < 0.9 * trunc)
        assert band.sum() > 10000
        err, ref = np.abs(gD - oD)[band], np.abs(pD - oD)[band]
        runs.append((np.percentile(err, q), np.percentile(ref, q)))
        print("band voxels", band.sum(), "GPU-vs-oracle p50/p90/p99:", runs[-1][0], "oracle-vs-reordered-oracle:", runs[-1][1])
        if run < 2:
            for o in (gi, gl):
                o.destroy()
    # the GPU may differ from the single-thread oracle by no more than another legal order of the same oracle
    # does (measured ratios 0.7-0.95; the exact comparison of this scan is tests/test_tsdf_deterministic_gpu.py's)
    med_err = np.median([r_[0] for r_ in runs], axis=0)
    med_ref = np.median([r_[1] for r_ in runs], axis=0)
    for j, p in enumerate(q):
        assert med_err[j] <= 1.5 * med_ref[j] + 0.01 * vs, (p, med_err, med_ref)
        assert all(r_[0][j] <= 2.0 * r_[1][j] + 0.02 * vs for r_ in runs), (p, runs)     # and no single run far out
    # the reconstructed wall x = +5 m (projective distance, near-normal rays)
    xs = (np.arange(lo[0], hi[0]) + 0.5) * vs
    sl = (slice(None), slice(40, 56), slice(30, 40))
    true = 5.0 - xs[:, None, None]
    sel = (np.abs(true) < 0.5 * trunc) & (gW[sl] > 0)
    assert np.abs(gD[sl] - true)[sel].max() < 0.75 * vs
    for o in (gi, gl):
        o.destroy()


def test_randomised_single_ray_sequences_are_bit_exact(capi, ctx):
    """six random integrator configurations (vps 8/16, voxel 5-30 cm, carving on/off, constant or
    1/z^2 weights, drop-off, sparsity compensation, low max_weight, freespace points), 60 rays
    each, one ray per scan: GPU == oracle in every voxel, bit for bit."""
    for seed in range(6):
        rng = np.random.default_rng(500 + seed)
        vps = 8 if seed % 2 else 16
        vs = float(rng.choice([0.05, 0.1, 0.2, 0.3]))
        kw = dict(default_truncation_distance=float(rng.uniform(2, 4) * vs),
                  max_ray_length_m=float(rng.uniform(20, 60) * vs),
                  min_ray_length_m=float(rng.uniform(0.5, 2) * vs),
                  voxel_carving_enabled=int(rng.integers(0, 2)), use_const_weight=int(rng.integers(0, 2)),
                  use_weight_dropoff=int(rng.integers(0, 2)),
                  use_sparsity_compensation_factor=int(rng.integers(0, 2)),
                  sparsity_compensation_factor=float(rng.uniform(1, 30)),
                  allow_clear=int(rng.integers(0, 2)), max_weight=float(rng.choice([3.0, 50.0, 10000.0])),
                  max_consecutive_ray_collisions=int(rng.integers(0, 4)),
                  start_voxel_subsampling_factor=float(rng.choice([1.0, 2.0, 4.0])))
        ocfg, gcfg = _both_cfg(capi, **kw)
        ol = orc.TsdfLayer(vs, vps)
        oi = orc.FastTsdfIntegrator(ocfg, ol)
        half = int(np.ceil(80 * vs / (vps * vs))) + 2
        gl = capi.TsdfLayer(ctx, vs, vps, (-half,) * 3, (2 * half,) * 3, 4096)
        gi = capi.FastTsdfIntegrator(ctx, gcfg, gl)
        for k in range(60):
            ax = rng.normal(0, 1, 3); ax /= np.linalg.norm(ax)
            ang = rng.uniform(-3, 3)
            T = np.r_[np.cos(ang / 2), np.sin(ang / 2) * ax, rng.uniform(-2, 2, 3) * vs * 5].astype(F)
            dirv = rng.normal(0, 1, 3); dirv /= np.linalg.norm(dirv)
            p = (dirv * rng.uniform(0.2, 70.0) * vs).astype(F)[None]
            col = rng.integers(0, 256, (1, 4)).astype(np.uint8)
            free = bool(rng.random() < 0.15)
            a = oi.integratePointCloud(T, p, col, free)
            b = gi.integratePointCloud(T, p, col, free)
            assert a == b, (seed, k, a, b)
        assert gl.stats()[1] == 0
        O, G = _as_dict(*ol.download(), vps), _as_dict(*gl.download(), vps)
        assert O.keys() == G.keys(), seed
        bad = [k for k in O if O[k] != G[k]]
        assert not bad, (seed, len(bad), bad[:2], [O[k] for k in bad[:2]], [G[k] for k in bad[:2]])
        for o in (gi, gl):
            o.destroy()


def test_rgbd_fullsize_scan_invariants(capi, ctx):
    """BASELINE config 4 size: one 640x480 depth image (307 200 points) at 0.05 m voxels,
    early-out disabled: the number of voxel updates, the allocated blocks and every voxel
    weight (integer ray counts) are order-independent and must equal the oracle's exactly."""
    vs, vps, trunc = 0.05, 16, 0.15
    kw = dict(default_truncation_distance=trunc, max_ray_length_m=5.0, use_const_weight=1,
              use_weight_dropoff=0, max_consecutive_ray_collisions=1 << 30)
    ocfg, gcfg = _both_cfg(capi, **kw)
    u, v = np.meshgrid((np.arange(640) - 319.5) / 525.0, (np.arange(480) - 239.5) / 525.0)
    d = np.stack([np.ones_like(u), -u, -v], -1).reshape(-1, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    lo, hi = np.array([-5.0, -4.0, -1.0]), np.array([5.0, 4.0, 3.0])
    origin = np.array([-2.0, 0.3, 0.4])
    with np.errstate(divide="ignore"):
        t = np.where(d > 0, (hi - origin) / d, (lo - origin) / d).min(1)
    pts = (d * t[:, None]).astype(F)
    assert len(pts) == 307200
    T = np.array([1, 0, 0, 0, *origin], F)
    # Which of several points that share a start cell -- or whose cells share a slot of the
    # 2^20-entry approximate set -- gets cast depends on the visiting order.  Keep one point
    # per slot (slot = (LongIndexHash(cell) + offset) & mask, offset = 1 on the first scan):
    # the surviving 10^4..10^5 rays are then cast by both sides, whatever the order.
    pg = (pts + origin.astype(F)).astype(F)
    cells = np.floor(pg * np.float32(2.0 / vs) + np.float32(1e-6)).astype(np.int64)
    h = (cells[:, 0] + cells[:, 1] * 17191 + cells[:, 2] * 17191 * 17191) & 0xFFFFFFFF
    slot = (h + 1) & ((1 << 20) - 1)
    _, first = np.unique(slot, return_index=True)
    pts = pts[np.sort(first)]
    assert len(pts) > 20000
    ol = orc.TsdfLayer(vs, vps)
    a = orc.FastTsdfIntegrator(ocfg, ol).integratePointCloud(T, pts)
    gl = capi.TsdfLayer(ctx, vs, vps, (-8, -6, -2), (16, 12, 7), 16 * 12 * 7)
    gi = capi.FastTsdfIntegrator(ctx, gcfg, gl)
    b = gi.integratePointCloud(T, pts)
    assert gl.stats()[1] == 0
    print("RGB-D full size:", len(pts), "rays, voxel updates oracle/gpu", a, b, "blocks", ol.num_blocks())
    assert a == b and a > 1_000_000
    lo_v, hi_v = (-128, -96, -32), (128, 96, 80)
    oD, oW, oA = _grids(*ol.download()[:3], vps, lo_v, hi_v)
    gD, gW, gA = _grids(*gl.download()[:3], vps, lo_v, hi_v)
    assert np.array_equal(oA, gA)
    assert np.array_equal(oW, gW)                    # integer ray counts: exact
    one = oW == 1                                    # crossed by exactly one ray: order independent, bit for bit
    assert one.sum() > 100 and np.array_equal(oD[one].view(np.uint32), gD[one].view(np.uint32)), one.sum()
    diff = np.abs(gD - oD)[oW > 0]
    assert np.percentile(diff, 99) < 1e-4 and diff.max() <= 2 * trunc
    for o in (gi, gl):
        o.destroy()


def test_degenerate_point_clouds(capi, ctx):
    """empty cloud, points at the sensor origin, too close, beyond the maximum range with and
    without allow_clear, freespace_points, and non-finite coordinates: no crash, no hang, and the
    finite cases equal the oracle bit for bit"""
    vs, vps = 0.1, 16
    T = np.array([1, 0, 0, 0, 0.05, -0.02, 0.03], F)
    for allow_clear in (0, 1):
        ocfg, gcfg = _both_cfg(capi, default_truncation_distance=0.3, max_ray_length_m=3.0, min_ray_length_m=0.2,
                               allow_clear=allow_clear, use_const_weight=0)
        ol = orc.TsdfLayer(vs, vps)
        oi = orc.FastTsdfIntegrator(ocfg, ol)
        gl = capi.TsdfLayer(ctx, vs, vps, (-4, -4, -4), (8, 8, 8), 512)
        gi = capi.FastTsdfIntegrator(ctx, gcfg, gl)
        scans = [np.zeros((0, 3), F),                                            # empty
                 np.zeros((1, 3), F),                                            # at the origin: ray length 0
                 np.array([[0.1, 0.0, 0.0]], F),                                 # closer than min_ray_length_m
                 np.array([[0.2, 0.0, 0.0]], F),                                 # exactly min_ray_length_m
                 np.array([[3.0, 0.0, 0.0]], F),                                 # exactly max_ray_length_m
                 np.array([[0.0, 3.5, 0.5]], F),                                 # beyond it: clearing ray or dropped
                 np.array([[1.0, 1.0, 0.0]], F)]                                 # an ordinary point, for contrast
        for k, pts in enumerate(scans):
            for freespace in (False, True):
                a = oi.integratePointCloud(T, pts, None, freespace)
                b = gi.integratePointCloud(T, pts, None, freespace)
                assert a == b, (allow_clear, k, freespace, a, b)
        obi, od, ow, oc = ol.download()
        gbi, gd, gw, gc = gl.download()
        assert set(map(tuple, obi)) == set(map(tuple, gbi)) and len(obi) > 0
        O, G = _as_dict(obi, od, ow, oc, vps), _as_dict(gbi, gd, gw, gc, vps)
        assert all(O[k][0] == G[k][0] and O[k][1] == G[k][1] for k in O)
        # non-finite coordinates never reach the layer (the reference would index with floor(NaN))
        before = gl.download()
        bad = np.array([[np.nan, 0, 1], [np.inf, 1, 1], [1, -np.inf, 0], [np.nan, np.nan, np.nan]], F)
        assert gi.integratePointCloud(T, bad, None) == 0
        after = gl.download()
        assert all(np.array_equal(x, y) for x, y in zip(before, after))
        for o in (gi, gl):
            o.destroy()


def test_layer_grows_with_the_sensor_and_never_drops(capi, ctx):
    """voxblox::Layer is unbounded; so is the GPU layer: created without a box or a pool size, it
    follows a sensor that drives 80 m while full LiDAR-shaped scans keep arriving back to back
    (no host synchronisation between scans).  Nothing may be dropped, and a scan integrated far
    from the start must look like the same scan integrated at the origin."""
    vs, vps = 0.2, 16
    _, gcfg = _both_cfg(capi, default_truncation_distance=0.6, max_ray_length_m=16.0, use_const_weight=1,
                        use_weight_dropoff=1, use_sparsity_compensation_factor=1,
                        sparsity_compensation_factor=20.0)
    az = np.linspace(-np.pi, np.pi, 512, endpoint=False)
    el = np.deg2rad(np.linspace(-16.6, 16.6, 32))
    A, E = np.meshgrid(az, el)
    dirs = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], -1).reshape(-1, 3)
    rng = np.random.default_rng(3)
    pts = (dirs * rng.uniform(2.0, 20.0, (len(dirs), 1))).astype(F)       # some returns beyond max range
    import torch
    dev = torch.from_numpy(pts).cuda()
    torch.cuda.synchronize()
    gl = capi.TsdfLayer(ctx, vs, vps)                                       # no box, no pool size
    gi = capi.FastTsdfIntegrator(ctx, gcfg, gl)
    n_scans = 40
    for k in range(n_scans):
        T = np.array([1, 0, 0, 0, 2.0 * k, 0.5 * k, 0.0], F)
        gi.integrate_device(T, dev.data_ptr(), None, len(pts))              # asynchronous
    n_blocks, dropped = gl.stats()
    assert dropped == 0 and gl.growths() >= 2
    bi, d, w, _ = gl.download()
    assert len(bi) == n_blocks > 1500
    # the corridor swept by the sensor is covered end to end
    assert bi[:, 0].min() * vs * vps < -15 and bi[:, 0].max() * vs * vps > 2.0 * (n_scans - 1) + 14
    # a single scan far away == the same scan at the origin, shifted by whole blocks
    far = capi.TsdfLayer(ctx, vs, vps)
    near = capi.TsdfLayer(ctx, vs, vps, (-8, -8, -8), (16, 16, 16), 4096)
    shift_blocks = np.array([25, -40, 3])
    for layer, t in ((near, np.zeros(3)), (far, shift_blocks * vs * vps)):
        integ = capi.FastTsdfIntegrator(ctx, gcfg, layer)
        n_upd = integ.integrate_device(np.array([1, 0, 0, 0, *t], F), dev.data_ptr(), None, 1, count=True)
        assert n_upd > 0
        integ.destroy()
    assert near.stats()[1] == 0 and far.stats()[1] == 0
    b1, d1, w1, c1 = far.download()
    A = {k: v for k, v in _as_dict(*near.download(), vps).items() if v[1] > 0}
    B = {k: v for k, v in _as_dict(b1 - shift_blocks, d1, w1, c1, vps).items() if v[1] > 0}
    # one ray: order-independent.  The far origin is not exactly representable in f32 voxel
    # coordinates, so a voxel the ray merely grazes may differ and distances agree to rounding.
    common = A.keys() & B.keys()
    assert len(common) >= len(A) - 2 and len(common) >= len(B) - 2 and len(common) > 20
    assert max(abs(A[k][0] - B[k][0]) for k in common) < 5e-4
    for o in (gi, gl, far, near):
        o.destroy()


def test_layer_upload_download_round_trip_and_resume(capi, ctx):
    """vgx_tsdf_layer_upload: a voxblox layer handed to the GPU integrator (a submap that already
    holds data) comes back bit for bit, and integration resumes on it as on the original."""
    vs, vps = 0.1, 16
    ocfg, gcfg = _both_cfg(capi, default_truncation_distance=0.3, max_ray_length_m=8.0)
    ol = orc.TsdfLayer(vs, vps)
    oi = orc.FastTsdfIntegrator(ocfg, ol)
    rng = np.random.default_rng(8)
    scans = [(np.array([1, 0, 0, 0, *rng.uniform(-1, 1, 3)], F), rng.uniform(-4, 4, (1, 3)).astype(F),
              rng.integers(0, 255, (1, 4)).astype(np.uint8)) for _ in range(60)]
    for T, p, c in scans[:30]:
        oi.integratePointCloud(T, p, c)
    bi, d, w, rgba = ol.download()
    gl = capi.TsdfLayer(ctx, vs, vps)
    gl.upload(bi, d, w, rgba)
    b2, d2, w2, c2 = gl.download()
    assert np.array_equal(b2, bi) and np.array_equal(d2, d) and np.array_equal(w2, w) and np.array_equal(c2, rgba)
    # resume: the GPU integrator's approximate sets start empty like a fresh voxblox integrator's, so
    # compare against a FRESH oracle integrator continuing on the oracle's layer
    oi2 = orc.FastTsdfIntegrator(ocfg, ol)
    gi = capi.FastTsdfIntegrator(ctx, gcfg, gl)
    for T, p, c in scans[30:]:
        oi2.integratePointCloud(T, p, c)
        gi.integratePointCloud(T, p, c)
    A = _as_dict(*ol.download(), vps)
    B = _as_dict(*gl.download(), vps)
    assert A.keys() == B.keys()
    bad = [k for k in A if not (A[k][0] == B[k][0] and A[k][1] == B[k][1] and A[k][2] == B[k][2])]
    assert not bad, bad[:5]
    gi.destroy()
    gl.destroy()
