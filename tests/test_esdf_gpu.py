"""ESDF generation on the device (voxgraph_submap.cpp:86) vs the CPU
restatement of voxblox's EsdfIntegrator, plus order-free properties of the result."""
import numpy as np
import pytest

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


def _dense(bi, a, vps, fill):
    lo = bi.min(0) * vps
    shape = (bi.max(0) - bi.min(0) + 1) * vps
    out = np.full(shape, fill, a.dtype)
    for b in range(len(bi)):
        o = bi[b] * vps - lo
        out[o[0]:o[0] + vps, o[1]:o[1] + vps, o[2]:o[2] + vps] = a[b].reshape(vps, vps, vps).transpose(2, 1, 0)
    return out


def _check_fixed_point(bi, tsdf_d, esdf_d, esdf_o, vs, vps, min_d=0.2, max_d=2.0, default=2.0):
    """every non-fixed observed voxel equals min(default, best 26-neighbour candidate)"""
    D = _dense(bi, esdf_d, vps, np.nan).astype(np.float64)
    O = _dense(bi, esdf_o, vps, 0).astype(bool)
    Tt = _dense(bi, tsdf_d, vps, np.nan)
    D[~O] = np.nan
    P = np.pad(D, 1, constant_values=np.nan)
    best = np.full(D.shape, np.inf)
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                nz = (dx != 0) + (dy != 0) + (dz != 0)
                if nz == 0:
                    continue
                N = P[1 + dx:1 + dx + D.shape[0], 1 + dy:1 + dy + D.shape[1], 1 + dz:1 + dz + D.shape[2]]
                with np.errstate(invalid="ignore"):
                    ok = (np.sign(N) == np.sign(D)) & (np.sign(D) != 0) & (np.abs(N) < max_d)
                cand = np.where(ok, np.abs(N) + np.float64(np.float32(np.sqrt(np.float32(nz))) * np.float32(vs)), np.inf)
                best = np.minimum(best, cand)
    free = O & ~(np.abs(Tt) < min_d)
    want = np.minimum(default, best)
    err = np.abs(np.abs(D) - want)[free]
    return float(err.max()), int(free.sum())


def test_esdf_matches_oracle_and_is_an_exact_fixed_point(capi, ctx):
    sm, _ = synth.config1_pair(asymmetric=True)          # 64^3, sphere(s) + ground
    g = capi.Submap(ctx, 0, sm.voxel_size, sm.vps, sm.block_index, sm.tsdf_distance, sm.tsdf_weight,
                    None, None)
    passes = g.generate_esdf()
    assert 1 <= passes <= 40
    _, _, ed, eo = g.download_layers(sm.vps)
    od, oo, n_upd = orc.esdf_from_tsdf(sm.voxel_size, sm.vps, sm.block_index, sm.tsdf_distance,
                                       sm.tsdf_weight)
    assert np.array_equal(eo, oo) and n_upd > 1000
    fixed = (sm.tsdf_weight >= 1e-6) & (np.abs(sm.tsdf_distance) < 0.2)
    assert np.array_equal(ed[fixed], sm.tsdf_distance[fixed])
    diff = np.abs(ed - od)[oo.astype(bool)]
    print("ESDF GPU vs oracle: max", diff.max(), "p99", np.percentile(diff, 99), "passes", passes)
    # the oracle (voxblox's queue) ignores improvements below min_diff_m = 1 mm; the GPU result is
    # the exact fixed point: equal to f32 rounding almost everywhere, within ~2 min_diff_m where
    # the queue stopped early, never above it (measured on LiDAR-built submaps:
    # profiles/r02_chain_compare_*.json, max 1.8 mm, p99 1e-7; on the synthetic 64^3 scene max 0.95 mm,
    # p99 0.2 mm)
    assert diff.max() < 2.5e-3 and np.percentile(diff, 99) < 1e-3 and np.all(np.abs(ed) <= np.abs(od) + 1e-6)
    err, n_free = _check_fixed_point(sm.block_index, sm.tsdf_distance, ed, eo, sm.voxel_size, sm.vps)
    assert n_free > 10000 and err < 1e-6, err
    # unobserved TSDF voxels stay unobserved; the sampling grid was rebuilt and is usable
    assert not eo[sm.tsdf_weight == 0].any()
    n = g.extract_voxel_points(1.0, 0.3, True)
    cf = capi.RegistrationCostFunction(ctx, g, g, capi.default_config(registration_point_type=capi.POINTS_VOXELS))
    r = np.ones(n)
    assert cf.Evaluate([np.zeros(4), np.zeros(4)], r, None) and np.all(r == 0)
    cf.destroy()
    g.destroy()


def test_esdf_sparse_blocks_and_sign_separation(capi, ctx):
    """dropped blocks stop the front; negative and positive sides never feed each other"""
    sdf = synth.sphere_ground_sdf((0.3, -0.2, 0.4), 0.9, -0.6)
    sm = synth.make_submap(sdf, 0.1, 16, (-2, -2, -2), (4, 4, 4), trunc=0.3, esdf_max=0.45,
                           drop_empty_blocks=True)
    assert sm.n_blocks < 64
    g = capi.Submap(ctx, 1, sm.voxel_size, sm.vps, sm.block_index, sm.tsdf_distance, sm.tsdf_weight, None, None)
    g.generate_esdf(capi.esdf_config(max_distance_m=1.0, default_distance_m=1.0, min_distance_m=0.15))
    _, _, ed, eo = g.download_layers(sm.vps)
    od, oo, _ = orc.esdf_from_tsdf(sm.voxel_size, sm.vps, sm.block_index, sm.tsdf_distance, sm.tsdf_weight,
                                   orc.esdf_config(max_distance_m=1.0, default_distance_m=1.0,
                                                   min_distance_m=0.15))
    assert np.array_equal(eo, oo)
    assert np.abs(ed - od)[oo.astype(bool)].max() < 2.5e-3
    obs = oo.astype(bool)
    assert np.array_equal(np.sign(ed[obs]), np.sign(sm.tsdf_distance[obs]))
    err, _ = _check_fixed_point(sm.block_index, sm.tsdf_distance, ed, eo, sm.voxel_size, sm.vps, 0.15, 1.0, 1.0)
    assert err < 1e-6
    g.destroy()


def test_esdf_fullsize_256_against_analytic_scene(capi, ctx):
    """256^3 city submap: the device ESDF from the TSDF agrees with the analytic distance
    field to within the quasi-Euclidean (26-neighbour) over-estimate."""
    pose = np.array([3.0, -2.0, 0.0, 0.1])
    g = capi.Submap.synth_city(ctx, 0, 0.2, 16, (-8, -8, -4), (16, 16, 16), 0.6, 2.0, 10.0, pose, 2)
    _, _, analytic, a_obs = g.download_layers(16)
    ctx.timer_start()
    passes = g.generate_esdf()
    ms = ctx.timer_stop()
    _, _, ed, eo = g.download_layers(16)
    print(f"ESDF 256^3: {passes} passes, {ms:.2f} ms")
    obs = eo.astype(bool)
    # the scene generator marks |d| <= 1.2 m as TSDF-observed; compare there, away from the cap
    sel = obs & (np.abs(analytic) < 1.0) & a_obs.astype(bool)
    over = np.abs(ed[sel]) - np.abs(analytic[sel])
    assert sel.sum() > 1_000_000
    assert np.array_equal(np.sign(ed[sel]), np.sign(analytic[sel]))
    assert over.min() > -0.15 and np.percentile(over, 99) < 0.2 and np.median(np.abs(over)) < 0.03
    g.destroy()


def test_the_max_distance_frontier_when_the_default_lies_beyond_it(capi, ctx):
    """Outside voxblox's defaults (max_distance_m = default_distance_m = 2 m) the two algorithms have ONE discontinuity: a voxel is
    reached only from a neighbour whose |distance| < max_distance_m, so where that neighbour lies within the queue's 1 mm slack
    (min_diff_m) of the limit, the exact fixed point propagates (max + a step) and the queue leaves the default -- or the other
    way round.  Found by profiles/fuzz_esdf.py (seed 5840: this scene); everything else agrees within 2.5 mm as everywhere."""
    rng = np.random.default_rng(5840)
    vps = int(rng.choice([8, 16])); vs = float(rng.choice([0.05, 0.1, 0.2]))
    dims = tuple(int(x) for x in rng.integers(1, 5, 3)); ext = np.array(dims) * vps * vs
    c = rng.uniform(0.2, 0.8, 3) * ext
    sdf = synth.sphere_ground_sdf(tuple(c), float(rng.uniform(0.2, 0.6) * ext.min()), float(rng.uniform(0.1, 0.4) * ext[2]))
    sm = synth.make_submap(sdf, vs, vps, tuple(int(x) for x in rng.integers(-2, 2, 3)), dims, trunc=3 * vs, esdf_max=10 * vs,
                           drop_empty_blocks=bool(rng.integers(0, 2)))
    td = sm.tsdf_distance.copy()
    if rng.integers(0, 2):
        td = np.clip(td + rng.normal(0, 0.02 * vs, td.shape).astype(F), -3 * vs, 3 * vs).astype(F)
    tw = sm.tsdf_weight.copy()
    if rng.integers(0, 2):
        tw = np.where(rng.uniform(size=tw.shape) < 0.03, 0, tw).astype(F)
    max_d = float(rng.choice([2.0, 6 * vs, 12 * vs]))
    kw = dict(max_distance_m=max_d, default_distance_m=float(rng.choice([max_d, 2.0])), min_distance_m=float(rng.choice([0.2, vs, 2 * vs])))
    assert kw["default_distance_m"] > kw["max_distance_m"]                       # the configuration the finding needs
    g = capi.Submap(ctx, 0, vs, vps, sm.block_index, td, tw, None, None)
    g.generate_esdf(capi.esdf_config(**kw))
    _, _, ed, eo = g.download_layers(vps)
    od, oo, _ = orc.esdf_from_tsdf(vs, vps, sm.block_index, td, tw, orc.esdf_config(**kw))
    assert np.array_equal(eo, oo)
    obs = oo.astype(bool)
    far = obs & (np.abs(ed - od) >= 2.5e-3)
    assert 1 <= int(far.sum()) <= 3                                              # (one voxel on this scene)
    step_max = float(np.float32(np.sqrt(np.float32(3.0))) * np.float32(vs))
    for d_dev, d_q in zip(np.abs(ed[far]), np.abs(od[far])):
        at_default, other = (d_q, d_dev) if d_q == F(kw["default_distance_m"]) else (d_dev, d_q)
        assert at_default == F(kw["default_distance_m"]) and kw["max_distance_m"] - 2.5e-3 < other <= kw["max_distance_m"] + step_max + 1e-6
    err, _ = _check_fixed_point(sm.block_index, td, ed, eo, vs, vps, kw["min_distance_m"], kw["max_distance_m"], kw["default_distance_m"])
    assert err < 1e-6                                                            # the device layer IS the recurrence's fixed point
    g.destroy()
